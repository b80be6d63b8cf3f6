"""Image batch assembly on the device: the per-iteration part of the reference's ``LLFFDataset`` (data/loader.py) -- ``__getitem__``
(:325-356) and ``unravel_idx_from_rayid`` (:118-123) on the resident ``images`` / ``poses`` (/ ``pts0_images``, set_pts0_prior :110-116).
Reading the dataset from disk (images, poses_bounds.npy) stays with the caller: on-disk formats are out of scope."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L

HALF_PIX = 0.5          # utils/rays.py:5


class ImageBatcher:
    """images [n_img, H, W, 3] float32, poses [n_img, 3, 4] float32 (the train split of loader.py:60-63), K [3, 3].
    ``batcher[ray_ids]`` returns the reference's dict: rays [n, 3, 2], rays_x / rays_y [n, 1] (pixel + 0.5), images_idx [n, 1] int64,
    rgbsf [n, 3], poses [n, 3, 4] (+ rgbsf_pts0 [n, 3] once a prior is set) -- one kernel launch."""

    def __init__(self, images, poses, K, device="cuda"):
        dev = torch.device(device)
        self.device = dev
        self.images = torch.as_tensor(images, dtype=torch.float32, device=dev).contiguous()
        self.poses = torch.as_tensor(poses, dtype=torch.float32, device=dev)[:, :3, :4].contiguous()
        if self.images.dim() != 4 or self.images.shape[-1] != 3 or self.poses.shape[0] != self.images.shape[0]:
            raise L.EvdError("ImageBatcher: images [n_img, H, W, 3] and one [3, 4] pose per image")
        self.n_imgs, self.h, self.w = (int(v) for v in self.images.shape[:3])
        self.n_rays = self.n_imgs * self.h * self.w
        self.K = np.ascontiguousarray(np.asarray(K.detach().cpu() if isinstance(K, torch.Tensor) else K, dtype=np.float32).reshape(-1))
        self.pts0_images = None
        self._invalid = torch.zeros((1,), dtype=torch.int32, device=dev)

    def __len__(self):
        return self.n_rays

    def set_pts0_prior(self, pts0_images):
        """loader.py:110-116"""
        p = torch.as_tensor(pts0_images, dtype=torch.float32, device=self.device).contiguous()
        assert p.shape[0] == self.images.shape[0]
        if tuple(p.shape) != tuple(self.images.shape):
            raise L.EvdError("set_pts0_prior: the prior images must have the shape of the images")
        self.pts0_images = p

    def unravel_idx_from_rayid(self, ray_id):
        """loader.py:118-123 (C order): ray id -> (image id, y, x)"""
        if isinstance(ray_id, torch.Tensor):
            i = torch.div(ray_id, self.h * self.w, rounding_mode="floor")
            rem = ray_id - i * (self.h * self.w)
            y = torch.div(rem, self.w, rounding_mode="floor")
            return i, y, rem - y * self.w
        return np.unravel_index(ray_id, (self.n_imgs, self.h, self.w), order="C")

    def __getitem__(self, ray_ids, check=False):
        """loader.py:325-356.  check=True reads the device flag back and raises IndexError for an id outside [0, len(self))."""
        if isinstance(ray_ids, int):
            ray_ids = [ray_ids]
        ids = torch.as_tensor(ray_ids, device=self.device).to(torch.int64).contiguous().reshape(-1)
        n, dev = ids.shape[0], self.device
        f32 = dict(dtype=torch.float32, device=dev)
        rays = torch.empty((n, 3, 2), **f32)
        rx, ry = torch.empty((n, 1), **f32), torch.empty((n, 1), **f32)
        idx = torch.empty((n, 1), dtype=torch.int64, device=dev)
        rgb = torch.empty((n, 3), **f32)
        poses = torch.empty((n, 3, 4), **f32)
        rgb0 = torch.empty((n, 3), **f32) if self.pts0_images is not None else None
        L.check(L.lib().evd_image_batch(L.ptr(ids), n, L.ptr(self.images), L.ptr(self.pts0_images), L.ptr(self.poses), self.n_imgs, self.h, self.w,
                                        self.K.ctypes.data_as(C.POINTER(C.c_float)), L.ptr(rays), L.ptr(rx), L.ptr(ry), L.ptr(idx), L.ptr(rgb),
                                        L.ptr(poses), L.ptr(rgb0), L.ptr(self._invalid), L.stream_ptr()), "evd_image_batch")
        if check and int(self._invalid.item()):
            raise IndexError("ImageBatcher: a ray id outside [0, n_imgs * h * w)")
        out = {"rays": rays, "rays_x": rx, "rays_y": ry, "images_idx": idx, "rgbsf": rgb, "poses": poses}
        if rgb0 is not None:
            out["rgbsf_pts0"] = rgb0
        return out
