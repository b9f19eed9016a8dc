"""`import _bind` first thing in a timing script: the process on the CPUs of GPU 0's NUMA node before the HIP runtime starts
(gaussianprocessnode_amd/hostbind.py says why)."""
import importlib.util, os
_spec = importlib.util.spec_from_file_location("_sgp_hostbind", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                             "gaussianprocessnode_amd", "hostbind.py"))
_hb = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_hb)
info = _hb.bind_to_gpu_node(int(os.environ.get("LOCAL_RANK", "0")))
