import sys, os, json, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_aux
from oxylus_amd.renderer import RendererInstance
args = types.SimpleNamespace(no_cpu_baseline=(len(sys.argv) > 1 and sys.argv[1] == "nocpu"), unordered_output=1)  # the main line's flags
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
r = RendererInstance(0); stream = torch.cuda.Stream(device=dev)
res = bench_aux.bench_real_geometry(args, r, dev, stream, 0)
print(json.dumps({k: res[k] for k in ("ms_per_frame", "bit_match", "kernels_avg_us", "visible_fraction", "roofline")}))
