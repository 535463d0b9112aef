import sys, os, torch, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.backends.cuda.preferred_blas_library(sys.argv[1])
sys.argv = ["train_epoch_bench.py", "--kind", sys.argv[2], "--epochs", "6"]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "train_epoch_bench.py"), run_name="__main__")
