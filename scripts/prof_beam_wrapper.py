import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.getcwd())
import openpystruct_amd as oa
import bench
dev = torch.device("cuda:0")
st = bench.synth_inputs(4, 0, dev, "trajectory")
out = oa.beam_solve(**st)
torch.cuda.synchronize()
def run():
    for _ in range(3000):
        oa.beam_solve(**st, out=out)
    torch.cuda.synchronize()
run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
