"""A C program (examples/c_abi_device.c: plain C99, gcc, the HIP runtime's C API — no Python, no torch, no C++) allocates device
memory and calls s360_layout / s360_forward / s360_backward on the same scene as the Python binding; every output must be
IDENTICAL byte for byte.  Exercises the device entry points' struct layout and argument order from the header's side — the
ctypes prototypes in splatter360_amd/_lib.py are written by the same hand as the kernels; this caller only sees include/s360.h
(the boundary a pybind / C++ binding of upstream's rasterize_gaussians[_backward] would use,
/root/reference/src/model/decoder/cuda_splatting.py:113-124)."""
import shutil
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

from splatter360_amd import _lib, decoder, rasterizer, synthetic

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _build(tmp_path):
    if shutil.which("gcc") is None or not Path("/opt/rocm/include/hip/hip_runtime_api.h").exists():
        pytest.skip("no gcc / HIP headers on this box")
    exe = tmp_path / "c_abi_device"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", f"-I{ROOT / 'include'}", "-I/opt/rocm/include", str(ROOT / "examples" / "c_abi_device.c"),
           f"-L{ROOT / 'splatter360_amd'}", "-ls360", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{ROOT / 'splatter360_amd'}",
           "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("lean", [True, False])
def test_c_program_reproduces_the_python_binding_bit_for_bit(gpu, tmp_path, lean):
    exe = _build(tmp_path)
    P, fw = 20_000, 96
    cloud = synthetic.uniform_cloud(P, seed=11, extent=3.0, scale_range=(0.02, 0.3))
    ps = [torch.tensor(cloud[k], device=gpu).requires_grad_(True) for k in ("means", "covariances", "harmonics", "opacities")]
    ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=gpu), 0.1, 10.0)
    views = decoder.pack_camera_views(ext, K, near, far, torch.tensor([0.05, 0.1, 0.15], device=gpu))
    cap = 400_000
    images, radii = rasterizer.rasterize_views(ps[0], ps[1], ps[3], ps[2], views=views, image_height=fw, image_width=fw, sh_degree=4,
                                               shared_campos=True, cov9=True, sh_channel_major=True, max_instances=cap, lean=lean)
    st = rasterizer.last_state()
    L = st.num_rendered()
    assert not st.overflowed() and L > 10_000
    dimg = torch.randn(images.shape, generator=torch.Generator().manual_seed(2)).to(gpu)
    images.backward(dimg)
    prm = st.prm
    assert not (prm.flags & _lib.FLAG_FORWARD_ONLY) and bool(prm.flags & _lib.FLAG_LEAN_LISTS) == lean
    scene, out = tmp_path / "scene.bin", tmp_path / "out.bin"
    with open(scene, "wb") as f:
        f.write(struct.pack("<9i", 0x53333630, prm.P, prm.V, prm.H, prm.W, prm.M, prm.sh_degree, prm.flags, prm.max_instances))
        for t in (views, ps[0], ps[1], ps[3], ps[2], dimg):
            f.write(t.detach().float().contiguous().cpu().numpy().tobytes())
    r = subprocess.run([str(exe), str(scene), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"num_instances {L} overflow 0" in r.stdout, r.stdout
    raw = out.read_bytes()
    off = 0

    def take(ref: torch.Tensor):
        nonlocal off
        a = ref.detach().contiguous().cpu().numpy()
        got = np.frombuffer(raw, dtype=a.dtype, count=a.size, offset=off).reshape(a.shape)
        off += a.nbytes
        return got, a

    for name, ref in (("images", images), ("radii", radii), ("d_means", ps[0].grad), ("d_covariances", ps[1].grad),
                      ("d_opacities", ps[3].grad), ("d_harmonics", ps[2].grad)):
        got, want = take(ref)
        assert np.array_equal(got, want), name
    head = np.frombuffer(raw, dtype=np.uint32, count=2, offset=off)
    assert int(head[0]) == L and int(head[1]) == 0 and off + 8 == len(raw)
