"""cProfile of a few FWI iterations of examples/marmousi_fwi.py (host-side overheads)."""
import cProfile, pstats, io, os, sys, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'examples'))
import marmousi_fwi as ex
args = ["--maxiter", "6", "--odir", "/tmp/prof_fwi"] + sys.argv[1:]
with contextlib.redirect_stdout(io.StringIO()):
    ex.main(["--maxiter", "1", "--odir", "/tmp/prof_fwi_warm"])
pr = cProfile.Profile()
pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    ex.main(args)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue())
