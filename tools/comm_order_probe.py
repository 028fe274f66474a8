"""Not a test: the RCCL communicator must come up whichever of libgypsum_hip / torch is loaded first."""
import os
import sys

sys.path.insert(0, os.getcwd())
order = sys.argv[1]
if order == "torch_first":
    import torch  # noqa: F401
from gypsum_amd.engine import GypsumEngine  # noqa: E402

eng = GypsumEngine(0)
if order == "gypsum_first":
    import torch  # noqa: F401
    import torch.distributed  # noqa: F401
uid = eng.comm_unique_id()
eng.comm_init(0, 1, uid)
os.system("grep -i 'rccl\\|amdhip\\|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
print(order, "comm ok", eng.comm_info(), flush=True)
