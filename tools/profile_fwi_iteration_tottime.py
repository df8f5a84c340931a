import cProfile, pstats, io, os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "devito-fwi_amd")]
import numpy as np, torch, bench
true_model, init_model = bench.marmousi2(16)
src_all, rec = bench.acquisition(true_model, 8)
for _ in range(2):
    print("fwi_iteration: %.2f ms" % bench.fwi_iteration(true_model, init_model, src_all, rec)[0], flush=True)
pr = cProfile.Profile(); pr.enable()
bench.fwi_iteration(true_model, init_model, src_all, rec, reps=5)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(40); print(s.getvalue())
