"""bench.py against another build of the library.  usage: python tools/r05/run_var.py <path/to/librnerf_variant.so> [bench args...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from samplenerfro_amd import _lib
_lib.load(os.path.join(ROOT, sys.argv[1]))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench
bench.main()
