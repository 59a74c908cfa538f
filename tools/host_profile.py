"""cProfile of the host side of the train step (where the ~11-16 ms of enqueue time per step go): python tools/host_profile.py [M|S] [nsteps]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import torch
from tensorflowasr_amd import configs
from tensorflowasr_amd.conformer import ConformerTransducer
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
b = importlib.util.module_from_spec(spec); argv = sys.argv; sys.argv = [argv[0]]; spec.loader.exec_module(b); sys.argv = argv
which = sys.argv[1] if len(sys.argv) > 1 else "M"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
cfg = configs.conformer_m() if which == "M" else configs.conformer_s()
m = ConformerTransducer(cfg, dev, dtype=torch.bfloat16, seed=0)
m.prefetched_inputs = True
data = b.to_train_data(b.make_batch(cfg, 32, seed=10, padding="batch", size="S-10s" if which == "S" else "LibriSpeech-shaped"), dev)
for _ in range(5):
    m.train_step(data)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(n):
    m.train_step(data)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"{which}: {1e3 * (t1 - t0) / n:.2f} ms of host time per step under the profiler ({n} steps)")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:]))
