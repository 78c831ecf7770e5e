"""Developer tool: run the C5 Vecchia-Laplace configuration once more than bench.py does, for a kernel trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
r = bench.vl_config(0)
print({k: v for k, v in r.items() if k != "what"})
