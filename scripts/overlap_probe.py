"""How well does an HBM-bound stream overlap with the latency-bound binning chain on this chip?  (gpurun)
A = fused forward alone, B = a 380-MB read alone, C = both back to back on one stream, D = the read on a side stream, forked
and joined with events around every forward."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from splatter360_amd import decoder, synthetic
dev = torch.device("cuda:0")
cloud = synthetic.encoder_like_cloud(512, 1024, seed=0)
ps = [torch.tensor(cloud[k], device=dev) for k in ("means", "covariances", "harmonics", "opacities")]
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
views = decoder.pack_camera_views(ext, K, near, far, torch.zeros(3, device=dev))
big = torch.randn(95_000_000, device=dev)      # 380 MB
side = torch.cuda.Stream()

def fwd():
    with torch.no_grad():
        return decoder.render_views_fused(ext, K, near, far, (256, 256), torch.zeros(3, device=dev), *ps, check="lazy", shared_campos=True, views=views)

def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6

def both_seq():
    fwd(); big.sum()

def both_par():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        s = big.sum()
        done = torch.cuda.Event(); done.record()
    fwd()
    torch.cuda.current_stream().wait_event(done)

print("A forward alone us", timeit(fwd))
print("B 380 MB read alone us", timeit(lambda: big.sum()))
print("C sequential us", timeit(both_seq))
print("D side stream (fork + join per step) us", timeit(both_par))
