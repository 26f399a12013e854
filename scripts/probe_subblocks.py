"""Would 4x4-pixel waves shorten the polar tiles' critical path?  For the longest tiles of the surface-like cloud: per 4x4 block,
the list length its pixels need (max n_contrib) and the entries that can reach it (alpha >= 1/255 on one of its 16 pixels),
against the same two numbers per 8x8 quadrant."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from splatter360_amd import decoder, rasterizer, synthetic
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "surface_like"
cloud = {"encoder_like": lambda: synthetic.encoder_like_cloud(512, 1024), "surface_like": lambda: synthetic.surface_like_cloud(512, 1024)}[name]()
g = [torch.tensor(cloud[k], device=dev).requires_grad_(True) for k in ("means", "covariances", "harmonics", "opacities")]
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
decoder.render_views_fused(ext, K, near, far, (256, 256), torch.zeros(3, device=dev), *g, shared_campos=True)
st = rasterizer.last_state()
t = st.tensors()
ts = t["tile_start"].cpu().numpy().astype(np.int64)
tl = ts[1:] - ts[:-1]
lst = t["list"].cpu().numpy().astype(np.int64)
P = g[0].shape[0]
ra = t["rec_a"].reshape(-1, 4).cpu().numpy().astype(np.float64)
rb = t["rec_b"].reshape(-1, 4).cpu().numpy().astype(np.float64)
nc = t["n_contrib"].cpu().numpy().astype(np.int64)
for tile in np.argsort(-tl)[:6]:
    v, rem = tile // 256, tile % 256
    ty, tx = rem // 16, rem % 16
    ent = lst[ts[tile]:ts[tile + 1]]
    x, y, a, b = ra[ent].T
    c, op = rb[ent, 0], rb[ent, 1]
    ncl = nc[v, 16 * ty:16 * ty + 16, 16 * tx:16 * tx + 16]
    ys, xs = np.mgrid[0:16, 0:16]
    dx = x[:, None, None] - (16 * tx + xs)[None]
    dy = y[:, None, None] - (16 * ty + ys)[None]
    power = a[:, None, None] * dx * dx + b[:, None, None] * dx * dy + c[:, None, None] * dy * dy
    reach = (op[:, None, None] * np.exp2(power) >= 1 / 255.0) & (power <= 0)          # [n, 16, 16]
    pos = np.arange(len(ent))
    out = []
    for bs in (8, 4):
        for by in range(16 // bs):
            for bx in range(16 // bs):
                blk = (slice(by * bs, by * bs + bs), slice(bx * bs, bx * bs + bs))
                rl = int(ncl[blk].max())
                hits = reach[:, blk[0], blk[1]].any(axis=(1, 2)) & (pos < rl)
                out.append((bs, by, bx, rl, int(hits.sum())))
    q = [o for o in out if o[0] == 8]
    s = [o for o in out if o[0] == 4]
    cost = lambda o: o[3] / 64 * 1.0 + o[4] * 0.1          # us: ~1 us per walked chunk, ~0.1 us per survivor (lone wave)
    print(f"tile {tile} len {tl[tile]}: 8x8 quadrants (replay, reaching): {[(o[3], o[4]) for o in q]}  est. chain {max(cost(o) for o in q):.0f} us")
    print(f"    4x4 blocks: max replay {max(o[3] for o in s)}, max reaching {max(o[4] for o in s)}, est. chain {max(cost(o) for o in s):.0f} us;  blocks: {[(o[3], o[4]) for o in s]}")
