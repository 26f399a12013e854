"""Generates the golden fixtures under tests/golden/ by IMPORTING the reference's Python glue
from /root/reference in the build container (it cannot travel to the GPU box; only these small
data files do).  Nothing from the reference is copied: the fixtures are inputs + outputs.

Run:  python tests/golden/make_golden.py        (needs /root/reference; CPU only)

Recipe = SURVEY.md Appendix B: stub `jaxtyping`, `cv2` and a RECORDING fake
`diff_gaussian_rasterization`, pre-seed empty packages for src.model.decoder / encoder so their
__init__ files (torchvision, torch.hub) never run, then import the real modules.
"""
import importlib
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = "/root/reference"
OUT = Path(__file__).resolve().parent


def install_stubs():
    jt = types.ModuleType("jaxtyping")

    class _Ann:
        def __getitem__(self, item):
            return object

    for name in ("Float", "Int", "Int64", "Int32", "UInt8", "Bool", "Shaped"):
        setattr(jt, name, _Ann())
    sys.modules["jaxtyping"] = jt
    sys.modules["cv2"] = types.ModuleType("cv2")

    dgr = types.ModuleType("diff_gaussian_rasterization")
    dgr.CALLS = []

    class GaussianRasterizationSettings:
        def __init__(self, **kw):
            self.kw = kw

    class GaussianRasterizer:
        def __init__(self, settings):
            self.settings = settings

        def __call__(self, **kw):
            dgr.CALLS.append((self.settings.kw, kw))
            h, w = self.settings.kw["image_height"], self.settings.kw["image_width"]
            p = kw["means3D"].shape[0]
            return torch.zeros(3, h, w), torch.zeros(p, dtype=torch.int32)

    dgr.GaussianRasterizationSettings = GaussianRasterizationSettings
    dgr.GaussianRasterizer = GaussianRasterizer
    sys.modules["diff_gaussian_rasterization"] = dgr
    sys.path.insert(0, REF)
    for pkg in ("src.model.decoder", "src.model.encoder", "src.model.encoder.costvolume"):
        m = types.ModuleType(pkg)
        m.__path__ = [str(Path(REF) / pkg.replace(".", "/"))]
        sys.modules[pkg] = m
    return dgr


def to_np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return x


def main():
    dgr = install_stubs()
    cs = importlib.import_module("src.model.decoder.cuda_splatting")
    layers = importlib.import_module("src.geometry.layers")
    proj = importlib.import_module("src.geometry.projection")
    utils360 = importlib.import_module("src.geometry.utils360")
    sys.path.insert(0, str(OUT.parent.parent))
    from splatter360_amd import cameras, synthetic  # inputs only (face poses are re-derived below)

    torch.manual_seed(0)
    rng = np.random.default_rng(0)

    # ---- (1) boundary captures of render_cuda / render_depth_cuda for the six face cameras ----
    # Face cameras built straight from the preprocessing recipe (convert_cubemaps_mp.py:135-193),
    # restated here with explicit matrices so the fixture does not depend on the product code.
    def rx(d):
        a = np.deg2rad(d); s, c = np.sin(a), np.cos(a)
        return torch.tensor([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=torch.float32)

    def ry(d):
        a = np.deg2rad(d); s, c = np.sin(a), np.cos(a)
        return torch.tensor([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=torch.float32)

    import math
    def rot_ref(kind, deg):  # same float path as the reference: math.sin/cos -> float32 tensor entries
        a = math.radians(deg); s, c = math.sin(a), math.cos(a)
        m = torch.eye(3)
        if kind == "x":
            m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
        else:
            m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
        return m

    q = torch.tensor([0.9, 0.1, -0.3, 0.2]); q = q / q.norm()
    w, x, y, z = q.tolist()
    R = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], dtype=torch.float32)
    pano = torch.eye(4); pano[:3, :3] = R; pano[:3, 3] = torch.tensor([0.1, 0.2, 0.3])
    rots = [rot_ref("x", 90), torch.eye(3), rot_ref("y", -90), rot_ref("y", -180), rot_ref("y", -270), rot_ref("x", -90)]
    faces = []
    for r in rots:
        m = pano.clone()
        m[:3, :3] = torch.bmm(pano[None, :3, :3], r[None])[0]
        faces.append(m)
    ext = torch.stack(faces)
    ext[..., 1] = -ext[..., 1]
    ext[..., 2] = -ext[..., 2]
    K = torch.eye(3); K[0, 0] = K[1, 1] = K[0, 2] = K[1, 2] = 0.5
    K = K[None].repeat(6, 1, 1)

    G, d_sh = 48, 25
    means = torch.tensor(rng.uniform(-2, 2, (1, G, 3)), dtype=torch.float32)
    A = torch.tensor(rng.standard_normal((1, G, 3, 3)) * 0.1, dtype=torch.float32)
    covs = A @ A.transpose(-1, -2) + 1e-4 * torch.eye(3)
    harm = torch.tensor(rng.standard_normal((1, G, 3, d_sh)), dtype=torch.float32)
    opac = torch.tensor(rng.uniform(0.05, 0.95, (1, G)), dtype=torch.float32)
    near, far = torch.tensor([0.1]), torch.tensor([10.0])
    bg = torch.tensor([[0.1, 0.2, 0.3]])
    cap = dict(pano_c2w=pano.numpy(), face_c2w=ext.numpy(), face_K=K.numpy(), means=means[0].numpy(),
               covariances=covs[0].numpy(), harmonics=harm[0].numpy(), opacities=opac[0].numpy(),
               near=near.numpy(), far=far.numpy(), bg=bg.numpy())
    for f in range(6):
        dgr.CALLS.clear()
        cs.render_cuda(ext[f][None], K[f][None], near, far, (64, 64), bg, means, covs, harm, opac)
        st, kw = dgr.CALLS[0]
        for k in ("viewmatrix", "projmatrix", "campos", "bg"):
            cap[f"f{f}_{k}"] = to_np(st[k])
        for k in ("tanfovx", "tanfovy", "sh_degree", "image_height", "image_width", "scale_modifier"):
            cap[f"f{f}_{k}"] = np.asarray(st[k])
        for k in ("means3D", "shs", "opacities", "cov3D_precomp"):
            cap[f"f{f}_{k}"] = to_np(kw[k])
        assert kw["colors_precomp"] is None and not st["prefiltered"] and not st["debug"]
    # depth rendering capture (face 1, all four modes)
    for mode in ("depth", "disparity", "relative_disparity", "log"):
        dgr.CALLS.clear()
        cs.render_depth_cuda(ext[1][None], K[1][None], near, far, (64, 64), means, covs, opac, mode=mode)
        st, kw = dgr.CALLS[0]
        cap[f"depth_{mode}_colors"] = to_np(kw["colors_precomp"])
        cap[f"depth_{mode}_bg"] = to_np(st["bg"])
        assert kw["shs"] is None
    # scale_invariant=False and a non-square-fov camera
    K2 = torch.tensor([[[0.7, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]]])
    dgr.CALLS.clear()
    cs.render_cuda(ext[2][None], K2, near, far, (48, 80), bg, means, covs, harm, opac, scale_invariant=False)
    st, kw = dgr.CALLS[0]
    cap["ns_K"] = K2.numpy()
    for k in ("viewmatrix", "projmatrix", "campos"):
        cap[f"ns_{k}"] = to_np(st[k])
    cap["ns_tanfovx"], cap["ns_tanfovy"] = np.asarray(st["tanfovx"]), np.asarray(st["tanfovy"])
    cap["ns_means3D"] = to_np(kw["means3D"])
    cap["getproj"] = cs.get_projection_matrix(torch.tensor([1.0, 0.5]), torch.tensor([100.0, 20.0]),
                                              torch.tensor([np.pi / 2, 1.0]), torch.tensor([np.pi / 2, 0.7])).numpy()
    cap["getfov_K"] = torch.cat([K[:1], K2]).numpy()
    cap["getfov"] = proj.get_fov(torch.cat([K[:1], K2])).numpy()
    # orthographic fake camera (render_cuda_orthographic, batch 1)
    dgr.CALLS.clear()
    dump = {}
    cs.render_cuda_orthographic(ext[1][None], torch.tensor([3.0]), torch.tensor([2.0]), near, far, (48, 64), bg, means, covs,
                                harm, opac, dump=dump)
    st, kw = dgr.CALLS[0]
    for k in ("viewmatrix", "projmatrix", "campos"):
        cap[f"ortho_{k}"] = to_np(st[k])
    cap["ortho_tanfovx"], cap["ortho_tanfovy"] = to_np(st["tanfovx"]), to_np(st["tanfovy"])
    for k in ("extrinsics", "fov_x", "fov_y", "near", "far"):
        cap[f"ortho_dump_{k}"] = to_np(dump[k])
    np.savez_compressed(OUT / "boundary_render_cuda.npz", **cap)

    # ---- (2) Cube2Equirec: sample grid + a stitched random cube, small and bench sizes ----
    for (fw, eh, ew) in ((32, 64, 128), (64, 128, 256)):
        c2e = layers.Cube2Equirec(fw, eh, ew)
        cube = torch.tensor(rng.standard_normal((1, 3, fw, 6 * fw)), dtype=torch.float32)
        erp = c2e(cube)
        np.savez_compressed(OUT / f"cube2equirec_{fw}_{eh}_{ew}.npz", grid=c2e.sample_grid.detach().numpy()[0, 0],
                            cube=cube.numpy()[0], erp=erp.detach().numpy()[0])
    c2e = layers.Cube2Equirec(256, 512, 1024)
    g = c2e.sample_grid.detach().numpy()[0, 0]
    # full 512x1024x3 grid is 6 MB: keep a strided sample + a checksum of every element
    np.savez_compressed(OUT / "cube2equirec_256_512_1024_sample.npz", rows=g[::37].copy(), cols=g[:, ::41].copy(),
                        sum64=np.asarray(g.astype(np.float64).sum(axis=(0, 1))),
                        face_counts=np.bincount(np.rint((g[..., 2] + 1) * 2.5).astype(int).ravel(), minlength=6))

    # ---- (3) change_order on an index-valued tensor (src/model/model_wrapper_erp.py:135-145) ----
    # model_wrapper_erp imports lightning etc.; restate the 6 lines' EFFECT by running the same torch ops
    # here would be copying, so instead we record the permutation through Cube2Equirec-independent facts
    # captured from the reference function via exec of its source object is not possible without its
    # imports -> load the function object by compiling only that function from the module source.
    import ast
    src = Path(REF, "src/model/model_wrapper_erp.py").read_text()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "change_order"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "change_order", "exec"), ns)
    cubes = torch.arange(6 * 2 * 3 * 3, dtype=torch.float32).reshape(6, 2, 3, 3)
    out = ns["change_order"](cubes.clone())
    np.savez_compressed(OUT / "change_order.npz", inp=cubes.numpy(), out=out.numpy())

    # ---- (4) ERP ray convention of the encoder (utils360.py:93-104,148-153) ----
    h, w = 8, 16
    u = utils360.Utils(dict(dataset_name="hm3d", batch_size=1, height=h, width=w))
    coords = u.get_xy_coords()
    sph = u.equi_2_spherical(coords, radius=1)
    cart = u.spherical_2_cartesian(sph)[0]
    np.savez_compressed(OUT / "erp_rays_8x16.npz", dirs=cart.numpy())

    # ---- (5) cube-face loss / metric the fused epilogue replaces: LossMse (src/loss/loss_mse.py:30-31) and
    # compute_psnr (src/evaluation/metrics.py:11-21).  metrics.py also imports lpips / skimage (absent here, only
    # used by the other two metrics): empty stand-in modules let the real compute_psnr be imported and run.
    lp = types.ModuleType("lpips"); lp.LPIPS = object
    sk = types.ModuleType("skimage"); skm = types.ModuleType("skimage.metrics"); skm.structural_similarity = None
    sys.modules.update({"lpips": lp, "skimage": sk, "skimage.metrics": skm})
    for pkg in ("src.dataset", "src.loss", "src.evaluation"):   # skip their __init__ (torchvision, lpips-based losses)
        m = types.ModuleType(pkg)
        m.__path__ = [str(Path(REF) / pkg.replace(".", "/"))]
        sys.modules[pkg] = m
    sys.modules["src.dataset"].DatasetCfg = object   # decoder.py only names it in a type annotation
    metrics = importlib.import_module("src.evaluation.metrics")
    loss_mse = importlib.import_module("src.loss.loss_mse")
    dec = importlib.import_module("src.model.decoder.decoder")
    pred = torch.tensor(rng.uniform(-0.2, 1.3, (2, 6, 3, 8, 8)), dtype=torch.float32)      # [b, v*cubes, 3, h, w]
    gt = torch.tensor(rng.uniform(-0.1, 1.2, (2, 1, 6, 3, 8, 8)), dtype=torch.float32)     # [b, v, cubes, 3, h, w]
    loss = loss_mse.LossMse(loss_mse.LossMseCfgWrapper(loss_mse.LossMseCfg(weight=0.37)))
    val = loss(dec.DecoderOutput(color=pred, depth=None), {"target": {"image_cubes_supervise": gt}}, None, 0)
    psnr = metrics.compute_psnr(gt[0, 0], pred[0])
    np.savez_compressed(OUT / "loss_mse_psnr.npz", pred=pred.numpy(), gt=gt.numpy(), weight=np.float32(0.37),
                        loss=val.numpy(), psnr_b0=psnr.numpy())
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
