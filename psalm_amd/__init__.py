import torch  # noqa: F401  (loads the HIP runtime that libpsalm_hip.so must share; see hip_ops.py)
