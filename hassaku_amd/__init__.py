

# RCCL / cross-process sharing of device memory needs dmabuf IPC on this driver stack; the variable is read when the
# HIP runtime initialises, i.e. it must be in the environment before the first torch.cuda call of the process
import os as _os
_os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
