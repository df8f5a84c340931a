"""GPU probe: BASELINE config 4 on one GPU (592^3, so=8, nt=500, one shot): forward + L2 gradient (bench.py's leg)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
import mifwi
import bench
print(json.dumps(bench.c4_leg(mifwi, 0, space_order=int(sys.argv[1]) if len(sys.argv) > 1 else 8), indent=1))
