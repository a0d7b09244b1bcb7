"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the video-frame transform of the reference's MSR-VTT loader.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The reference (src/data/__init__.py:22-26) composes, for a clip of uint8 frames [T, 3, H, W]:
    torchvision ConvertImageDtype(float32)      -> x / 255                       (fp32 division)
    torchvision Resize((S, S), BICUBIC, antialias=True) on a TENSOR
        -> torch.nn.functional.interpolate(mode="bicubic", align_corners=False, antialias=True)
           = ATen upsample_bicubic2d_aa (third-party: PyTorch; the reference pins no version, 2.10.0 is the one here)
    torchvision Normalize(mean, std)            -> (x - mean) / std

ATen's antialiased resize is a separable convolution in float: for every output index i
    scale = in / out, support = 2 * max(scale, 1), center = scale * (i + 0.5),
    taps xmin = max(int(center - support + 0.5), 0) .. min(int(center + support + 0.5), in) - 1,
    weight_j = cubic((j - center + 0.5) / max(scale, 1)) with a = -0.5, normalised to sum 1,
horizontal pass first, no clamping of the float result (over/undershoot is kept).  This file restates it in
numpy float32.  Pinned by tests/test_video_formats.py against torch's own CPU operator on the same inputs (torch is a
dependency of the reference and is importable wherever the tests run); float arithmetic: tolerance 2e-6 on the [0, 1]
scale, stated in the tests.
"""
import numpy as np

f32 = np.float32


def _cubic(x):
    """ATen bicubic_aa_filter, a = -0.5, evaluated in float32."""
    a = f32(-0.5)
    x = np.abs(x.astype(f32))
    near = ((a + f32(2)) * x - (a + f32(3))) * x * x + f32(1)
    far = (((x - f32(5)) * x + f32(8)) * x - f32(4)) * a
    return np.where(x < 1, near, np.where(x < 2, far, f32(0))).astype(f32)


def aa_coeffs(in_size, out_size):
    """-> (xmin int[out], xsize int[out], weights float32[out, kmax]) as ATen computes them for one axis."""
    scale = f32(in_size) / f32(out_size)
    support = f32(2.0) * scale if scale >= 1 else f32(2.0)
    invscale = f32(1.0) / scale if scale >= 1 else f32(1.0)
    kmax = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int64)
    xsize = np.zeros(out_size, np.int64)
    w = np.zeros((out_size, kmax), f32)
    for i in range(out_size):
        center = f32(np.float64(scale) * (i + 0.5))
        lo = max(int(np.float64(center - support) + 0.5), 0)
        hi = min(int(np.float64(center + support) + 0.5), in_size)
        n = hi - lo
        j = np.arange(n)
        arg = (((j + lo).astype(f32) - center).astype(np.float64) + 0.5) * np.float64(invscale)
        wt = _cubic(arg.astype(f32))
        total = f32(0)
        for v in wt:
            total = f32(total + v)
        if total != 0:
            wt = (wt / total).astype(f32)
        xmin[i], xsize[i] = lo, n
        w[i, :n] = wt
    return xmin, xsize, w


def _resize_axis(x, out_size, axis):
    """x float32 [..., in, ...] -> [..., out, ...]; sequential float32 accumulation in tap order."""
    x = np.moveaxis(x, axis, -1)
    xmin, xsize, w = aa_coeffs(x.shape[-1], out_size)
    out = np.empty(x.shape[:-1] + (out_size,), f32)
    for i in range(out_size):
        acc = x[..., xmin[i]] * w[i, 0]
        for j in range(1, xsize[i]):
            acc = (acc + x[..., xmin[i] + j] * w[i, j]).astype(f32)
        out[..., i] = acc
    return np.moveaxis(out, -1, axis)


def resize_bicubic_aa(x, size):
    """x float32 [..., H, W] -> [..., S, S]: horizontal pass, then vertical pass."""
    x = np.ascontiguousarray(x, dtype=f32)
    if x.shape[-1] != size:
        x = _resize_axis(x, size, -1)
    if x.shape[-2] != size:
        x = _resize_axis(x, size, -2)
    return x


def preprocess_frames(frames_u8, size, mean, std):
    """frames uint8 [T, H, W, 3] (decoder layout) -> float32 [T, 3, S, S]."""
    x = np.transpose(np.asarray(frames_u8), (0, 3, 1, 2)).astype(f32) / f32(255.0)
    x = resize_bicubic_aa(x, size)
    m = np.asarray(mean, f32).reshape(1, 3, 1, 1)
    s = np.asarray(std, f32).reshape(1, 3, 1, 1)
    return ((x - m) / s).astype(f32)
