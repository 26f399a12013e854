import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from splatter360_amd import decoder, rasterizer, stitch, synthetic, cameras
dev = torch.device("cuda:0")
cloud = synthetic.encoder_like_cloud(512, 1024)
params = [torch.tensor(cloud[k], device=dev, requires_grad=True) for k in ("means", "covariances", "harmonics", "opacities")]
pose = torch.eye(4, device=dev)
ext, K, near, far = decoder.cube_cameras(pose, 0.1, 10.0)
bg = torch.zeros(3, device=dev)
gt = torch.full((6, 3, 256, 256), 0.5, device=dev)
c2e = stitch.Cube2Equirec(256, 512, 1024).to(dev)
def timeit(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e3*(t1-t0)/n:.3f} ms/iter, total {1e3*(t2-t0)/n:.3f} ms/iter", flush=True)
timeit("pack_camera_views", lambda: decoder.pack_camera_views(ext, K, near, far, bg))
views = decoder.pack_camera_views(ext, K, near, far, bg)
def fwd_only():
    with torch.no_grad():
        return rasterizer.rasterize_views(params[0], params[1], params[3], params[2], None, views=views, image_height=256, image_width=256, sh_degree=4, shared_campos=True, want_radii=False, check="lazy", cov9=True, sh_channel_major=True)
timeit("rasterize fwd (no grad)", fwd_only)
def fwdbwd():
    for p in params: p.grad = None
    f, _ = rasterizer.rasterize_views(params[0], params[1], params[3], params[2], None, views=views, image_height=256, image_width=256, sh_degree=4, shared_campos=True, want_radii=False, check="lazy", cov9=True, sh_channel_major=True)
    ((f - gt) ** 2).mean().backward()
timeit("rasterize fwd+bwd", fwdbwd)
def full():
    for p in params: p.grad = None
    f = decoder.render_views_fused(ext, K, near, far, (256, 256), bg, *params, check="lazy")
    e = c2e.stitch_rendered(f.detach())
    ((f - gt) ** 2).mean().backward()
timeit("full step", full)
cams = decoder.CameraPrefetcher(dev)
def fwd_step():
    with torch.no_grad():
        v = cams.pack(ext, K, near, far, bg)
        f = decoder.render_views_fused(ext, K, near, far, (256, 256), bg, *[p.detach() for p in params], check="lazy", views=v)
        return c2e.stitch_rendered(f)
timeit("bench fwd step (prefetcher)", fwd_step)
def fwd_step_noglue():
    with torch.no_grad():
        f = decoder.render_views_fused(ext, K, near, far, (256, 256), bg, *[p.detach() for p in params], check="lazy", views=views)
        return c2e.stitch_rendered(f)
timeit("bench fwd step (views cached)", fwd_step_noglue)
