"""Headless Canvas (ref: Canvas.py): camera matrices and PNG output on the CPU; the GPU rasteriser against a NumPy
restatement of transform / draw_sphere / draw_point / fill_pixel (Canvas.py:138-205) with a nearest-fragment rule."""
import struct
import zlib

import numpy as np
import pytest

from wcsph_amd.Canvas import Canvas


def test_static_cam_matrices():
    c = Canvas(512, 256)
    c.static_cam(0.0, 1.0, 0.0)                       # dfsph.py:604
    assert c.ortho == 1 and c.fov == 2.0 and c.scale == 3.0
    np.testing.assert_allclose(c.eye, [0.0, 1.0, 3.0])
    np.testing.assert_allclose(c.view, [[1, 0, 0, 0], [0, 1, 0, -1], [0, 0, 1, -3], [0, 0, 0, 1]], atol=1e-7)
    ys = 1.0 / np.tan(1.0)
    np.testing.assert_allclose(c.proj, [[ys / 2, 0, 0, 0], [0, ys, 0, 0], [0, 0, 1 / (1 - 1000.0), 1 / (1 - 1000.0)], [0, 0, 0, 1]], rtol=1e-6)
    c.yaw_cam(0.0, 1.0, 0.0)                          # perspective orbit: yaw advances by 0.003 per call
    assert c.ortho == 0 and abs(c.yaw - 0.003) < 1e-12
    assert c.proj[3, 2] == -1.0


def test_png_writer_roundtrip(tmp_path):
    c = Canvas(4, 3)
    c.out_dir = str(tmp_path)
    a = c.img.to_numpy()
    a[0, 0] = (1, 0, 0)          # bottom-left pixel red
    a[3, 2] = (0, 0, 1)          # top-right pixel blue
    c.export_png(0.0)
    assert c.frame == 1
    c.export_png(0.0)            # same frame time: nothing written twice (Canvas.py:125-129)
    assert c.frame == 1
    data = open(tmp_path / "0.png", "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xffffffff
        chunks[tag] = body
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    assert (w, h, depth, ctype) == (4, 3, 8, 2)
    raw = zlib.decompress(chunks[b"IDAT"])
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all()
    px = rows[:, 1:].reshape(h, w, 3)
    assert tuple(px[2, 0]) == (255, 0, 0) and tuple(px[0, 3]) == (0, 0, 255)       # y up: row 0 is the top


def reference_image(pos, n_liquid, pv, sx, sy):
    """serial restatement with `depth > z` replaced by nearest-fragment (ties: smaller packed colour)"""
    depth = np.full((sx, sy), 1.0, np.float32)
    img = np.zeros((sx, sy, 3), np.float32)
    white, grey = np.float32(1.0), np.float32(round(0.3 * 1023) / 1023)
    p = np.concatenate([pos, np.ones((len(pos), 1), np.float32)], 1).astype(np.float32) @ pv.T.astype(np.float32)
    p = p / p[:, 3:4]
    fx = (p[:, 0] + 1) * np.float32(0.5) * np.float32(sx); fy = (p[:, 1] + 1) * np.float32(0.5) * np.float32(sy)

    def fill(x, y, z, c):
        if 0 <= x < sx and 0 <= y < sy and (z < depth[x, y] or (z == depth[x, y] and z < 1.0 and c < img[x, y, 0])):
            depth[x, y] = z; img[x, y] = c
    for i in range(len(pos)):
        xc, yc, z = int(fx[i]), int(fy[i]), p[i, 2]
        if i < n_liquid:
            x, y, d = 0, 3, 3 - 2 * 3
            while x <= y:
                for ax, ay in ((x, y), (-x, y), (x, -y), (-x, -y), (y, x), (-y, x), (y, -x), (-y, -x)):
                    fill(xc + ax, yc + ay, z, white)
                if d < 0:
                    d = d + 4 * x + 6
                else:
                    d = d + 4 * (x - y) + 10; y -= 1
                x += 1
        fill(xc, yc, z, grey)
    return img


@pytest.mark.gpu
@pytest.mark.parametrize("cam", ["static", "yaw"])
def test_gpu_frame_matches_restatement(cam, tmp_path):
    from util import GpuSim, small_scene
    liquid, solid = small_scene(6, jitter=0.3)
    g = GpuSim(liquid, solid, mode="clean")
    c = Canvas(256, 192)
    if cam == "static":
        c.static_cam(0.0, 1.0, 0.0)
    else:
        c.yaw = 0.6
        c.yaw_cam(0.0, 1.0, 0.0)
    c.clear_canvas()
    c.draw_particle(g.pd)
    got = c.img.to_numpy()
    want = reference_image(g.pd.pos.to_numpy(), len(liquid), c.proj @ c.view, 256, 192)
    assert (want.sum(axis=2) > 0).sum() > 500                     # the scene is on screen
    diff = np.abs(got - want).max(axis=2) > 2e-3
    assert diff.sum() <= 0.01 * (want.sum(axis=2) > 0).sum(), diff.sum()      # float-rounding of a pixel coordinate at most
    c.out_dir = str(tmp_path)
    c.export_png_frame(3)
    assert (tmp_path / "3.png").exists() and c.frame == 3
    g.close()


@pytest.mark.gpu
def test_driver_writes_frames_and_meshes(tmp_path):
    """the reference's main loop body, headless: PNG frames (dfsph.py:643) and marching-cubes meshes (:635)"""
    from wcsph_amd import dfsph
    dfsph.main(steps=3, particleDim=6, png_dir=str(tmp_path), surface_dir=str(tmp_path))
    assert (tmp_path / "0.png").exists()
    mesh = (tmp_path / "mc_0.obj").read_text().splitlines()
    assert sum(l.startswith("f ") for l in mesh) > 100
    dfsph.particle_data.close()
