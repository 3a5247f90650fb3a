#!/bin/bash
# A/B libraries for the micro-benchmarks' --libs / --lib options: libpsalm_hip.so with ONE translation unit taken from another revision (or built
# with extra -D flags), everything else from the current build (psalm_amd/lib/obj/*.o -- run `python -m psalm_amd.build` first).
#   tools/experiments/build_side_lib.sh <name> <git-rev|WORKTREE> <unit> [extra hipcc flags]
#   e.g.  build_side_lib.sh r05head_winattn f7a93b9 attention            (the window / mha kernels before DESIGN section 0 item 7a)
#         build_side_lib.sh prepipe 4cb1748 gemm                         (gemm.hip before the epilogue prefetch)
#         build_side_lib.sh klds1024 WORKTREE attention -DPSALM_WINATTN_KLDS_MAX=1024
# -> tools/experiments/_build/libpsalm_hip_<name>.so (git-ignored; travels to the GPU box with the snapshot)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; REV=$2; UNIT=$3; shift 3
TMP=$(mktemp -d)
mkdir -p "$TMP/csrc" "$ROOT/tools/experiments/_build"
cp "$ROOT"/psalm_amd/csrc/*.h "$TMP/csrc/"
if [ "$REV" = "WORKTREE" ]; then
  cp "$ROOT/psalm_amd/csrc/$UNIT.hip" "$TMP/csrc/"
else
  for h in common.h; do git -C "$ROOT" show "$REV:psalm_amd/csrc/$h" > "$TMP/csrc/$h" 2>/dev/null || true; done
  git -C "$ROOT" show "$REV:psalm_amd/csrc/$UNIT.hip" > "$TMP/csrc/$UNIT.hip"
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I "$ROOT/include" "$@" -c "$TMP/csrc/$UNIT.hip" -o "$TMP/$UNIT.o"
OBJS=$(ls "$ROOT"/psalm_amd/lib/obj/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/experiments/_build/libpsalm_hip_$NAME.so" "$TMP/$UNIT.o" $OBJS
rm -rf "$TMP"
echo "$ROOT/tools/experiments/_build/libpsalm_hip_$NAME.so"
