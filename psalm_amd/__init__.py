import os

import torch  # noqa: F401  (loads the HIP runtime that libpsalm_hip.so must share; see hip_ops.py)

if os.environ.get("PSALM_AMD_DROPIN", "0") not in ("", "0"):     # PSALM_AMD_DROPIN=1: `import psalm_amd` installs the drop-in (dropin.py)
    from . import dropin as _dropin
    _dropin.install()
