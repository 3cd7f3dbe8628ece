"""Test helper: the HIP kernels (through the C-ABI, kintinuous_amd/abi.py) behind the numpy-in / numpy-out function names of
oracle/oracle.py and oracle/ref.py, so that one scenario can be run through the oracle, the reference build (oracle/_ref) and the
HIP path and compared.  Every call uploads its inputs, launches, downloads: the arithmetic is the product's, the plumbing is not."""
import numpy as np


class HipKernels:
    def __init__(self, ctx):
        from kintinuous_amd import abi
        self.c = ctx
        self.abi = abi

    def _intr(self, i):
        return self.abi.Intr(i.fx, i.fy, i.cx, i.cy)

    # ---- image side ----
    def bilateral_filter(self, src):
        src = np.ascontiguousarray(src, np.uint16)
        rows, cols = src.shape
        dst = self.c.zeros(src.nbytes)
        self.c.bilateral_filter(self.c.upload(src), dst, cols, rows)
        return self.c.download(dst, np.uint16, src.shape)

    def pyr_down(self, src):
        src = np.ascontiguousarray(src, np.uint16)
        rows, cols = src.shape
        dst = self.c.zeros((rows // 2) * (cols // 2) * 2)
        self.c.pyr_down(self.c.upload(src), cols, rows, dst)
        return self.c.download(dst, np.uint16, (rows // 2, cols // 2))

    def create_vmap(self, intr, depth):
        depth = np.ascontiguousarray(depth, np.uint16)
        rows, cols = depth.shape
        v = self.c.zeros(3 * rows * cols * 4)
        self.c.create_vmap(self._intr(intr), self.c.upload(depth), cols, rows, v)
        return self.c.download(v, np.float32, (3 * rows, cols))

    def create_nmap(self, vmap):
        vmap = np.ascontiguousarray(vmap, np.float32)
        rows, cols = vmap.shape[0] // 3, vmap.shape[1]
        n = self.c.zeros(vmap.nbytes)
        self.c.create_nmap(self.c.upload(vmap), cols, rows, n)
        return self.c.download(n, np.float32, vmap.shape)

    def transform_maps(self, vmap, nmap, R, t):
        vmap, nmap = np.ascontiguousarray(vmap, np.float32), np.ascontiguousarray(nmap, np.float32)
        rows, cols = vmap.shape[0] // 3, vmap.shape[1]
        vd, nd = self.c.zeros(vmap.nbytes), self.c.zeros(vmap.nbytes)
        self.c.transform_maps(self.c.upload(vmap), self.c.upload(nmap), cols, rows, R, t, vd, nd)
        return self.c.download(vd, np.float32, vmap.shape), self.c.download(nd, np.float32, vmap.shape)

    def resize_map(self, inp, normalize):
        inp = np.ascontiguousarray(inp, np.float32)
        rows, cols = inp.shape[0] // 3, inp.shape[1]
        o = self.c.zeros(3 * (rows // 2) * (cols // 2) * 4)
        (self.c.resize_nmap if normalize else self.c.resize_vmap)(self.c.upload(inp), cols, rows, o)
        return self.c.download(o, np.float32, (3 * (rows // 2), cols // 2))

    def depth_to_metres(self, src, cutoff):
        src = np.ascontiguousarray(src, np.uint16)
        rows, cols = src.shape
        d = self.c.zeros(rows * cols * 4)
        self.c.depth_to_metres(self.c.upload(src), d, cols, rows, cutoff)
        return self.c.download(d, np.float32, src.shape)

    def bgr_to_intensity(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        rows, cols = rgb.shape[:2]
        d = self.c.zeros(rows * cols)
        self.c.bgr_to_intensity(self.c.upload(rgb), d, cols, rows)
        return self.c.download(d, np.uint8, (rows, cols))

    def pyr_down_gauss_f32(self, src):
        src = np.ascontiguousarray(src, np.float32)
        rows, cols = src.shape
        d = self.c.zeros((rows // 2) * (cols // 2) * 4)
        self.c.pyr_down_gauss_f32(self.c.upload(src), cols, rows, d)
        return self.c.download(d, np.float32, (rows // 2, cols // 2))

    def pyr_down_gauss_u8(self, src):
        src = np.ascontiguousarray(src, np.uint8)
        rows, cols = src.shape
        d = self.c.zeros((rows // 2) * (cols // 2))
        self.c.pyr_down_gauss_u8(self.c.upload(src), cols, rows, d)
        return self.c.download(d, np.uint8, (rows // 2, cols // 2))

    def derivative_images(self, src):
        src = np.ascontiguousarray(src, np.uint8)
        rows, cols = src.shape
        dx, dy = self.c.zeros(rows * cols * 2), self.c.zeros(rows * cols * 2)
        self.c.derivative_images(self.c.upload(src), cols, rows, dx, dy)
        return self.c.download(dx, np.int16, src.shape), self.c.download(dy, np.int16, src.shape)

    def project_to_cloud(self, depth, fx, fy, cx, cy, level):
        depth = np.ascontiguousarray(depth, np.float32)
        rows, cols = depth.shape
        cl = self.c.zeros(rows * cols * 12)
        self.c.project_to_cloud(self.c.upload(depth), cols, rows, cl, fx, fy, cx, cy, level)
        return self.c.download(cl, np.float32, (rows, cols, 3))

    # ---- tracking ----
    def icp_step(self, Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev, dist_thres, angle_thres):
        rows, cols = vmap_curr.shape[0] // 3, vmap_curr.shape[1]
        up = lambda a: self.c.upload(np.ascontiguousarray(a, np.float32))
        return self.c.icp_step(Rcurr, tcurr, up(vmap_curr), up(nmap_curr), Rprev_inv, tprev, self._intr(intr), up(vmap_g_prev), up(nmap_g_prev),
                               cols, rows, dist_thres, angle_thres)

    def rgb_residual(self, min_scale, dIdx, dIdy, last_depth, next_depth, last_image, next_image, max_depth_delta, kt, krkinv):
        rows, cols = next_image.shape
        cor = self.c.zeros(rows * cols * 16)
        up = self.c.upload
        s, n = self.c.rgb_residual(float(min_scale), up(np.ascontiguousarray(dIdx, np.int16)), up(np.ascontiguousarray(dIdy, np.int16)),
                                   up(np.ascontiguousarray(last_depth, np.float32)), up(np.ascontiguousarray(next_depth, np.float32)),
                                   up(np.ascontiguousarray(last_image, np.uint8)), up(np.ascontiguousarray(next_image, np.uint8)), cols, rows, cor,
                                   float(np.float32(max_depth_delta)), np.asarray(kt, np.float32), np.asarray(krkinv, np.float32))
        return self.c.download(cor, self.abi.DATATERM_DTYPE, (rows, cols)), s, n

    def rgb_step(self, corres, sigma, cloud, fx, fy, dIdx, dIdy, sobel_scale):
        rows, cols = dIdx.shape
        up = self.c.upload
        return self.c.rgb_step(up(np.ascontiguousarray(corres)), float(sigma), up(np.ascontiguousarray(cloud, np.float32)), float(fx), float(fy),
                               up(np.ascontiguousarray(dIdx, np.int16)), up(np.ascontiguousarray(dIdy, np.int16)), float(sobel_scale), cols, rows)

    # ---- volume ----
    def integrate_tsdf(self, depth, intr, volume_size, Rcurr_inv, tcurr, tranc_dist, volume, voxel_wrap, color_volume, colors, nmap_curr, angle_color):
        """In place on the numpy volume / colour volume (like the oracle's); returns depth_scaled."""
        depth = np.ascontiguousarray(depth, np.uint16)
        rows, cols = depth.shape
        N = volume.shape[0]
        dv, dc = self.c.upload(volume), self.c.upload(color_volume)
        sc = self.c.zeros(rows * cols * 4)
        self.c.integrate_tsdf(self.c.upload(depth), cols, rows, self._intr(intr), volume_size, Rcurr_inv, tcurr, tranc_dist, dv, sc, voxel_wrap, dc,
                              self.c.upload(np.ascontiguousarray(colors, np.uint8)), self.c.upload(np.ascontiguousarray(nmap_curr, np.float32)),
                              angle_color, N)
        self.c.sync()
        volume[...] = self.c.download(dv, np.int16, volume.shape)
        color_volume[...] = self.c.download(dc, np.uint8, color_volume.shape)
        return self.c.download(sc, np.float32, (rows, cols))

    def raycast(self, intr, Rcurr, tcurr, tranc_dist, volume_size, volume, vmap, nmap, voxel_wrap, vmap_color, color_volume):
        rows, cols = vmap.shape[0] // 3, vmap.shape[1]
        N = volume.shape[0]
        dvm, dnm, dcm = self.c.upload(vmap), self.c.upload(nmap), self.c.upload(vmap_color)
        self.c.raycast(self._intr(intr), Rcurr, tcurr, tranc_dist, volume_size, self.c.upload(volume), dvm, dnm, cols, rows, voxel_wrap, dcm,
                       self.c.upload(color_volume), N)
        self.c.sync()
        vmap[...] = self.c.download(dvm, np.float32, vmap.shape)
        nmap[...] = self.c.download(dnm, np.float32, nmap.shape)
        vmap_color[...] = self.c.download(dcm, np.uint8, vmap_color.shape)

    def clear_volume(self, vol, axis, back, current_wrap, delta_wrap):
        N = vol.shape[0]
        d = self.c.upload(vol)
        self.c.clear_volume(d, 2 if vol.dtype == np.int16 else 4, N, axis, int(back), current_wrap, delta_wrap)
        self.c.sync()
        vol[...] = self.c.download(d, vol.dtype, vol.shape)

    def extract_cloud_slice(self, volume, volume_size, cap, voxel_wrap, color_volume, minX, maxX, minY, maxY, minZ, maxZ, subsample, real_wrap):
        N = volume.shape[0]
        out = self.c.zeros(cap * 32)
        n = self.c.extract_cloud_slice(self.c.upload(volume), volume_size, out, cap, voxel_wrap, self.c.upload(color_volume), minX, maxX, minY, maxY,
                                       minZ, maxZ, subsample, real_wrap, N)
        return self.c.download(out, self.abi.POINT_DTYPE, (cap,))[:n]
