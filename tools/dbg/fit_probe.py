"""Model.fit at the C2 shapes: samples/s of the API users call vs bench.py's step."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
dev = torch.device("cuda")
model, schema = bench.build_model(dev)
model.compile(optimizer="adagrad", learning_rate=0.01)
B = 65536
raw = [bench.make_batch(dev, B, i) for i in range(8)]
split = lambda t: ({k: v for k, v in t.items() if k != "__label__"}, t["__label__"])
data = [split(raw[i % 8]) for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300)]
model(data[0][0])
for mode in (None, False):
    h = model.fit(data, graph=mode)
    print("fit graph=%s: %.1f M samples/s  (%.3f ms/step)" % (mode, h["examples_per_sec"][0] / 1e6, B / h["examples_per_sec"][0] * 1e3))
