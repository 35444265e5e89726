"""2-D image resampling between the full frame and a camera viewport.

`crop_boxes` = Camera.zoom / crop_to_viewport (reference modules/geometry.py:20-44,287-354):
grid_sample with zeros padding over a box; `uncrop` = Camera.uncrop (:261-285): paste a crop
back into the frame with border padding.  Both run once per observation (pre-processing) or
on (N,1,H,W) maps in the pose loss.

Device tensors go through `lf_grid_sample2d_*` (csrc/sample2d.hip); the pose loop's uncrop is fused into
the HIP loss kernels (engine.py).  Host tensors (an Observation that has not been moved to the GPU yet,
the CPU test-suite) are resampled by ATen on the host: that is host-side pre-processing, not a fallback
of the device path -- every device tensor takes the HIP kernel or raises.
"""
import torch
import torch.nn.functional as F

from . import ops


def _sample(image, grid, mode, padding):
    if image.is_cuda:
        return ops.grid_sample2d(image, grid, mode, padding)
    return F.grid_sample(image.float(), grid.float(), mode=mode, padding_mode=padding, align_corners=False)


def crop_boxes(image, boxes, target_size, scale_mode):
    """image (N,C,H,W), boxes (N,4) in frame pixels -> (N,C,T,T).

    SURVEY/oracle quirk Q15: the reference's TorchScript `bbox_to_grid` truncates the box
    corners toward zero to integers (aten::Int) before building the crop grid."""
    N, _, H, W = image.shape
    corners = boxes.detach().cpu().to(torch.float64).trunc().tolist()
    grids = []
    for x0, y0, x1, y1 in corners:            # same fp32 linspace arithmetic as the reference, box by box
        gy = torch.linspace(y0 / H, y1 / H, target_size, device=image.device) * 2 - 1
        gx = torch.linspace(x0 / W, x1 / W, target_size, device=image.device) * 2 - 1
        grids.append(torch.stack((gx[None, :].expand(target_size, -1), gy[:, None].expand(-1, target_size)), dim=-1))
    grid = torch.stack(grids, dim=0)
    return _sample(image, grid, scale_mode, 'zeros')


def uncrop(image, viewport, height, width, scale_mode):
    """image (N,C,h,w) living in `viewport` (N,4) -> (N,C,height,width), border replicated."""
    dev = image.device
    yy = torch.arange(0, height, device=dev, dtype=torch.float32).view(1, height, 1)
    xx = torch.arange(0, width, device=dev, dtype=torch.float32).view(1, 1, width)
    vh = (viewport[:, 3] - viewport[:, 1]).view(-1, 1, 1)
    vw = (viewport[:, 2] - viewport[:, 0]).view(-1, 1, 1)
    gy = ((yy - viewport[:, 1].view(-1, 1, 1)) / vh * 2 - 1).expand(-1, -1, width)
    gx = ((xx - viewport[:, 0].view(-1, 1, 1)) / vw * 2 - 1).expand(-1, height, -1)
    return _sample(image, torch.stack((gx, gy), dim=-1), scale_mode, 'border')
