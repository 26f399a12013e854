import sys, time, threading
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from splatter360_amd import decoder, rasterizer, synthetic
dev = torch.device("cuda:0")
c = synthetic.surface_like_cloud(1024, 2048)
ps0 = [torch.tensor(c[k], device=dev) for k in ("means", "covariances", "harmonics", "opacities")]
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
bg = torch.zeros(3, device=dev); gt = torch.full((6, 3, 512, 512), 0.5, device=dev)
def run():
    ps = [p.clone().requires_grad_(True) for p in ps0]
    faces, fm = decoder.render_views_fused(ext, K, near, far, (512, 512), bg, *ps, shared_campos=True, mse_target=gt, split_lists=True)
    st = rasterizer.last_state(); fm.loss.backward(); t = st.tensors()
    return [faces.detach().clone(), t["final_T"].clone(), t["n_contrib"].clone()] + [p.grad for p in ps], st
ref, st = run(); torch.cuda.synchronize()
print("split quadrants", int(st.header()[5].item()), "items", int(st.header()[6].item()), "err", st.split_errors())
# a second stream that keeps the chip busy with GEMMs while the split composite runs
stop = False
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
def hog():
    with torch.cuda.stream(side):
        while not stop:
            for _ in range(20): torch.mm(a, b)
            side.synchronize()
for phase in ("alone", "with GEMMs on a second stream"):
    if phase != "alone":
        th = threading.Thread(target=hog); th.start()
    bad = 0; t0 = time.time()
    for i in range(150):
        out, st = run()
        bad += not all(torch.equal(x, y) for x, y in zip(ref, out))
        assert st.split_errors() == 0, hex(st.split_errors())
    torch.cuda.synchronize()
    print(phase, "150 steps", round(time.time() - t0, 1), "s; differing:", bad)
stop = True; th.join()
