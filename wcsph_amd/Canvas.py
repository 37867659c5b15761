"""Canvas: the reference's viewer canvas without a window (ref: Canvas.py).  Same camera state and methods; the
particles are rasterised on the GPU (csrc/canvas.h) and frames are written as PNG files (no GUI, no taichi).

    sph_canvas = Canvas(512, 512); sph_canvas.static_cam(0.0, 1.0, 0.0)
    sph_canvas.clear_canvas(); sph_canvas.draw_particle(particle_data); sph_canvas.export_png(current_time)
"""
import math
import struct
import zlib

import numpy as np

from . import _lib


class _Image:
    def __init__(self, sx, sy):
        self._a = np.zeros((sx, sy, 3), np.float32)

    def to_numpy(self):
        return self._a


class Canvas:
    def __init__(self, sizex, sizey):
        self.view = np.eye(4, dtype=np.float32)
        self.proj = np.eye(4, dtype=np.float32)
        self.screenRes = (sizex, sizey)
        self.img = _Image(sizex, sizey)
        self.eye = np.array([0.0, 0.0, 1.0])
        self.target = np.array([0.0, 0.0, 0.0])
        self.up = np.array([0.0, 1.0, 0.0])
        self.ratio = sizex / sizey
        self.yaw = 0.0
        self.pitch = 0.0
        self.roll = 0.0
        self.scale = 1.0
        self.fov = 1.0
        self.near = 1.0
        self.far = 1000.0
        self.ortho = 0
        self.frame = 0
        self.fps = 30.0
        self.out_dir = "."

    # ---- camera (ref: Canvas.py:38-122; same numbers, python doubles stored as float32) -------------------------
    def _aim(self, target, fov, ortho):
        self.fov, self.ortho = fov, ortho
        self.target[:] = target

    def yaw_cam(self, targetx, targety, targetz):
        """perspective orbit about the vertical axis, 0.003 rad per call up to half a turn"""
        self._aim((targetx, targety, targetz), 1.0, 0)
        if self.yaw < 3.14:
            self.set_view_point(self.yaw + 0.003, 0.0, 0.0, 3.0)

    def pitch_cam(self, targetx, targety, targetz):
        """perspective tilt, 0.003 rad per call up to 0.5 rad"""
        self._aim((targetx, targety, targetz), 1.0, 0)
        if self.pitch < 0.5:
            self.set_view_point(0.0, self.pitch + 0.003, 0.0, 3.0)

    def static_cam(self, targetx, targety, targetz):
        """orthographic front view from distance 3 (the camera dfsph.py:604 uses)"""
        self._aim((targetx, targety, targetz), 2.0, 1)
        self.set_view_point(0.0, 0.0, 0.0, 3.0)

    def update_cam(self):
        self.pitch = float(np.clip(self.pitch, -1.57, 1.57))
        cp, sp, cy, sy = math.cos(self.pitch), math.sin(self.pitch), math.cos(self.yaw), math.sin(self.yaw)
        self.eye[:] = self.target + self.scale * np.array([cp * sy, sp, cp * cy])
        self.up[:] = (-sp * sy, cp, -sp * cy)
        # right-handed look-at: rows are the camera axes, the last column moves the eye to the origin
        back = self.eye - self.target
        back /= np.linalg.norm(back)
        right = np.cross(self.up, back)
        right /= np.linalg.norm(right)
        axes = np.stack([right, np.cross(back, right), back])
        view = np.eye(4)
        view[:3, :3] = axes
        view[:3, 3] = -axes @ self.eye
        self.view = view.astype(np.float32)
        ys = 1.0 / math.tan(self.fov / 2.0)
        n, f = self.near, self.far
        proj = np.zeros((4, 4))
        proj[0, 0], proj[1, 1] = ys / self.ratio, ys
        if self.ortho == 0:
            proj[2, 2], proj[2, 3], proj[3, 2] = f / (n - f), n * f / (n - f), -1.0
        else:
            proj[2, 2], proj[2, 3], proj[3, 3] = 1.0 / (n - f), n / (n - f), 1.0
        self.proj = proj.astype(np.float32)

    def set_view_point(self, yaw, pitch, roll, scale):
        self.yaw, self.pitch, self.roll, self.scale = yaw, pitch, roll, scale
        self.update_cam()

    def set_fov(self, fov):
        self.fov = fov
        self.update_cam()

    def set_target(self, targetx, targety, targetz):
        self.target[:] = (targetx, targety, targetz)
        self.update_cam()

    # ---- drawing ---------------------------------------------------------------------------------------------------
    def clear_canvas(self):
        """ref: Canvas.py:207-211"""
        self.img._a[...] = 0.0

    def draw_particle(self, particle_data, color_sphere=(1.0, 1.0, 1.0), color_point=(0.3, 0.3, 0.3)):
        """ref: dfsph.py:585-593 (clears first: the device resolves the whole frame in one call)"""
        ctx = particle_data._need_ctx()
        pv = np.ascontiguousarray((self.proj @ self.view).astype(np.float32))
        cs = np.asarray(color_sphere, np.float32); cp = np.asarray(color_point, np.float32)
        sx, sy = self.screenRes
        _lib.check(ctx, _lib.load().wcsph_canvas_draw(ctx, pv.ctypes.data, sx, sy, cs.ctypes.data, cp.ctypes.data, self.img._a.ctypes.data))

    # ---- output: ref Canvas.py:125-135 (ti.imwrite: img[x, y], y up) --------------------------------------------
    def _write(self, name):
        a = self.img.to_numpy()
        rgb = (np.clip(a, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
        rows = np.ascontiguousarray(np.transpose(rgb, (1, 0, 2))[::-1])     # [row from top][x][3]
        h, w = rows.shape[:2]
        raw = b"".join(b"\x00" + rows[r].tobytes() for r in range(h))

        def chunk(tag, data):
            return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
        png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
        import os
        path = os.path.join(self.out_dir, name)
        with open(path, "wb") as f:
            f.write(png)
        return path

    def export_png(self, time):
        time_i = int(time * self.fps)
        if int(time_i) == self.frame:
            self._write(str(self.frame) + ".png")
            self.frame += 1

    def export_png_frame(self, frame):
        if frame > self.frame:
            self.frame = frame
            self._write(str(self.frame) + ".png")
