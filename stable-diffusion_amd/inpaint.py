"""Pre-processing of the inpainting job on the device: the reference's `pad_image`, `make_batch_sd` and the c_cat loop of
`inpaint()` (LatentDiffusionInpainting.py:58-89, :279-293), same names and argument meaning, with the arithmetic in two
small HIP kernels (sdmi_inpaint_prepare, sdmi_posterior_sample) instead of numpy / torch host code.  Bit-identical to the
host code: every value is exact fp32 arithmetic on uint8 pixels (tests/test_gpu_round2.py)."""
import numpy as np
import torch

from . import _lib as L
from . import engine as E

SCALE_FACTOR = 0.18215


def padded_size(w, h):
    """(w, h) -> the size pad_image() pads to: next multiple of 64, at least 128 (LatentDiffusionInpainting.py:59-60)."""
    pw, ph = np.max(((2, 2), np.ceil(np.array((w, h)) / 64).astype(int)), axis=0) * 64
    return int(pw), int(ph)


def pad_image(input_image):
    """The reference's pad_image (LatentDiffusionInpainting.py:58-63): PIL image in, PIL image out, bottom / right edge
    replication.  Host code, for callers that want the padded PIL image itself; make_batch_sd() below does the padding
    on the device and does not need it."""
    from PIL import Image
    pw, ph = padded_size(*input_image.size)
    return Image.fromarray(np.pad(np.array(input_image), ((0, ph - input_image.size[1]), (0, pw - input_image.size[0]), (0, 0)), mode='edge'))


def _u8(img, what):
    a = np.asarray(img)
    if a.dtype != np.uint8 or a.ndim not in (2, 3):
        raise L.SdmiError("%s: expected a uint8 image (H, W) or (H, W, C), got %s %s" % (what, a.dtype, a.shape))
    return a if a.ndim == 3 else a[:, :, None]


def make_batch_sd(image, mask, txt, num_samples=1, device="cuda", pad=True, with_latent_mask=None):
    """make_batch_sd (LatentDiffusionInpainting.py:65-89) on the device.  image / mask: PIL images or uint8 arrays (H, W, 3) /
    (H, W[, 1 | 3]), UNPADDED when pad=True (the edge padding of pad_image is applied inside the kernel) or already padded.
    Returns the reference's dict (fp32 cuda tensors); with_latent_mask: an optional (num_samples, C, H/8, W/8) fp32 tensor
    whose channel 0 receives the nearest-resized mask (the mask branch of inpaint()'s c_cat loop, :283-285)."""
    img = _u8(image.convert("RGB") if hasattr(image, "convert") else image, "make_batch_sd image")
    msk = _u8(mask, "make_batch_sd mask") if not hasattr(mask, "convert") else _u8(mask.convert("RGB"), "make_batch_sd mask")
    if img.shape[2] != 3 or msk.shape[2] not in (1, 3) or img.shape[:2] != msk.shape[:2]:
        raise L.SdmiError("make_batch_sd: image %s / mask %s" % (img.shape, msk.shape))
    h0, w0 = img.shape[:2]
    w, h = padded_size(w0, h0) if pad else (w0, h0)
    dev = torch.device(device)
    di = torch.from_numpy(np.array(img, copy=True)).to(dev)
    dm = torch.from_numpy(np.array(msk, copy=True)).to(dev)
    n = num_samples
    out = {"image": torch.empty((n, 3, h, w), dtype=torch.float32, device=dev),
           "txt": n * [txt],
           "mask": torch.empty((n, 1, h, w), dtype=torch.float32, device=dev),
           "masked_image": torch.empty((n, 3, h, w), dtype=torch.float32, device=dev)}
    ml, mls = 0, 0
    if with_latent_mask is not None:
        t = with_latent_mask
        if tuple(t.shape[0:1] + t.shape[2:]) != (n, h // 8, w // 8) or t.dtype != torch.float32 or not t.is_contiguous():
            raise L.SdmiError("make_batch_sd: with_latent_mask must be contiguous fp32 (%d, C, %d, %d)" % (n, h // 8, w // 8))
        ml, mls = t.data_ptr(), t.shape[1] * t.shape[2] * t.shape[3]
    with torch.cuda.device(dev):
        L.call("sdmi_inpaint_prepare", di.data_ptr(), dm.data_ptr(), h0, w0, msk.shape[2], h, w, n, out["image"].data_ptr(),
               out["mask"].data_ptr(), out["masked_image"].data_ptr(), ml, mls, E._stream())
        torch.cuda.current_stream().synchronize()      # di / dm die with this frame
    return out


def image_conditioning(vae, batch, noise=None, c_concat=None):
    """The c_cat loop of inpaint() (LatentDiffusionInpainting.py:279-293): (n, 5, H/8, W/8) = [nearest-resized mask |
    0.18215 * encode(masked_image).sample()].  noise: the posterior's N(0, 1) draw, (n, 4, H/8, W/8) (drawn here when
    None).  The mask channel is taken from batch["mask"] (pixel (8i, 8j), as F.interpolate's default does) unless the
    caller passed `c_concat` already holding it (make_batch_sd(with_latent_mask=c_concat))."""
    mask, masked = batch["mask"], batch["masked_image"]
    n, _, h, w = masked.shape
    dev = masked.device
    if c_concat is None:
        c_concat = torch.empty((n, 5, h // 8, w // 8), dtype=torch.float32, device=dev)
        c_concat[:, 0] = mask[:, 0, ::8, ::8]
    moments = vae.encode(masked).parameters.contiguous()
    if noise is None:
        noise = torch.randn((n, 4, h // 8, w // 8), device=dev)
    noise = noise.contiguous().float()
    hw = (h // 8) * (w // 8)
    L.call("sdmi_posterior_sample", moments.data_ptr(), noise.data_ptr(), SCALE_FACTOR, c_concat.data_ptr() + 4 * hw,
           5 * hw, n, hw, E._stream())
    return c_concat
