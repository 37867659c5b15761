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

    # ---- camera: ref Canvas.py:38-122, unchanged arithmetic (python doubles, stored as f32) ---------------------
    def yaw_cam(self, targetx, targety, targetz):
        self.fov = 1.0
        self.ortho = 0
        self.target[:] = (targetx, targety, targetz)
        if self.yaw < 3.14:
            self.set_view_point(self.yaw + 0.003, 0.0, 0.0, 3.0)

    def pitch_cam(self, targetx, targety, targetz):
        self.fov = 1.0
        self.ortho = 0
        self.target[:] = (targetx, targety, targetz)
        if self.pitch < 0.5:
            self.set_view_point(0.0, self.pitch + 0.003, 0.0, 3.0)

    def static_cam(self, targetx, targety, targetz):
        self.fov = 2.0
        self.ortho = 1
        self.target[:] = (targetx, targety, targetz)
        self.set_view_point(0.0, 0.0, 0.0, 3.0)

    def update_cam(self):
        self.pitch = max(min(self.pitch, 1.57), -1.57)
        self.eye[0] = self.target[0] + self.scale * math.cos(self.pitch) * math.sin(self.yaw)
        self.eye[1] = self.target[1] + self.scale * math.sin(self.pitch)
        self.eye[2] = self.target[2] + self.scale * math.cos(self.pitch) * math.cos(self.yaw)
        self.up[0] = -math.sin(self.pitch) * math.sin(self.yaw)
        self.up[1] = math.cos(self.pitch)
        self.up[2] = -math.sin(self.pitch) * math.cos(self.yaw)
        zaxis = self.eye - self.target
        zaxis = zaxis / np.linalg.norm(zaxis)
        xaxis = np.cross(self.up, zaxis)
        xaxis = xaxis / np.linalg.norm(xaxis)
        yaxis = np.cross(zaxis, xaxis)
        yScale = 1.0 / math.tan(self.fov / 2.0)
        xScale = yScale / self.ratio
        self.view = np.array([[xaxis[0], xaxis[1], xaxis[2], -np.dot(xaxis, self.eye)],
                              [yaxis[0], yaxis[1], yaxis[2], -np.dot(yaxis, self.eye)],
                              [zaxis[0], zaxis[1], zaxis[2], -np.dot(zaxis, self.eye)],
                              [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)
        n, f = self.near, self.far
        if self.ortho == 0:
            self.proj = np.array([[xScale, 0, 0, 0], [0, yScale, 0, 0], [0, 0, f / (n - f), n * f / (n - f)], [0, 0, -1.0, 0]], dtype=np.float32)
        else:
            self.proj = np.array([[xScale, 0, 0, 0], [0, yScale, 0, 0], [0, 0, 1.0 / (n - f), n / (n - f)], [0, 0, 0, 1.0]], dtype=np.float32)

    def set_view_point(self, yaw, pitch, roll, scale):
        self.pitch, self.yaw, self.roll, self.scale = pitch, yaw, roll, scale
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
