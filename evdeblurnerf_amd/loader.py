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


# ------------------------------------------------------------------------------------------ once-per-dataset pose preparation (host, numpy)
def load_poses(poses_bounds, factor, imgshape, bdsmin=None, bd_factor=.75, scale=None):
    """LLFFDataset.load_poses (data/loader.py:178-203) on the array of poses_bounds.npy [N, 17] (reading the file stays with the caller):
    image size and the focal of the down-scaled images into column 4, LLFF column change [r1, -r0, r2, t, hwf], float32, translations and
    bounds rescaled by 1 / (min bound x bd_factor) (or `scale`).  -> poses [N, 3, 5] float32, bds [N, 2] float32, sc.  A few dozen 3 x 5
    matrices: host arithmetic, as in the reference."""
    poses_arr = np.array(poses_bounds, dtype=np.float64, copy=True)
    poses = poses_arr[:, :-2].reshape([-1, 3, 5])
    if not _is_pure_rotation_matrix(poses[:, :3, :3]):
        raise L.EvdError("load_poses: poses_bounds does not hold rotation matrices (the reference asserts, loader.py:181)")
    bds = poses_arr[:, -2:]
    poses[:, :2, 4] = np.array(imgshape[:2]).reshape([1, 2])
    poses[:, 2, 4] = poses[:, 2, 4] * 1. / factor
    poses = np.concatenate([poses[..., 1:2], -poses[..., 0:1], poses[..., 2:]], -1).astype(np.float32)
    bds = bds.astype(np.float32)
    bdsmin = np.min(bds) if bdsmin is None else bdsmin
    sc = (1. if bd_factor is None else 1. / (bdsmin * bd_factor)) if scale is None else scale
    poses[:, :3, 3] *= sc
    bds *= sc
    return poses, bds, sc


def _is_pure_rotation_matrix(M):
    """utils/data.py:9-31: every det isclose to 1 (numpy's default tolerances) and M^T allclose to inv(M) with atol 5e-7"""
    M = np.asarray(M)
    if not np.all(np.isclose(np.linalg.det(M), 1.0)):
        return False
    return bool(np.allclose(np.transpose(M, (0, 2, 1)), np.linalg.inv(M), atol=5e-7))


def _normalize(x):
    return x / np.linalg.norm(x)


def poses_avg(poses):
    """utils/data.py poses_avg: centre = mean translation, z = normalised sum of the z axes, up = sum of the y axes -> the view matrix with
    the hwf column of pose 0 ([3, 5])"""
    hwf = poses[0, :3, -1:]
    center = poses[:, :3, 3].mean(0)
    vec2 = _normalize(_normalize(poses[:, :3, 2].sum(0)))          # (poses_avg normalises, viewmatrix normalises again: utils/data.py:119-134)
    up = poses[:, :3, 1].sum(0)
    vec0 = _normalize(np.cross(up, vec2))
    vec1 = _normalize(np.cross(vec2, vec0))
    return np.concatenate([np.stack([vec0, vec1, vec2, center], 1), hwf], 1)


def recenter_poses(poses, c2w=None, return_c2w=False):
    """utils/data.py:167-183: inv(average pose) @ poses on the [3, 4] part; c2w: a given average pose (the event loader re-applies the image
    dataset's, run_nerf.py:74-81) -> poses (same dtype), and the [4, 4] c2w with return_c2w"""
    poses_ = poses + 0
    bottom = np.reshape([0, 0, 0, 1.], [1, 4])
    if c2w is None:
        c2w = poses_avg(poses)
        c2w = np.concatenate([c2w[:3, :4], bottom], -2)
    bottom = np.tile(np.reshape(bottom, [1, 1, 4]), [poses.shape[0], 1, 1])
    p44 = np.concatenate([poses[:, :3, :4], bottom], -2)
    p44 = np.linalg.inv(c2w) @ p44
    poses_[:, :3, :4] = p44[:, :3, :4]
    return (poses_, c2w) if return_c2w else poses_


def image_batcher_from_llff_arrays(poses_bounds, images, K=None, factor=1, bd_factor=.75, recenter=True, recenter_partial=None, device="cuda"):
    """The pose side of LLFFDataset.__init__ (data/loader.py:50-63 through load_poses :178-203 and the recentring branch of
    recenter_spherify_poses :205-216) on arrays: -> (ImageBatcher over the recentred [N, 3, 4] poses, dict(poses [N, 3, 5], bds, sc,
    recenter_partial [4, 4] -- what EventTables / PoseTrack re-apply)).  K None: load_intrinsics (:133-137) from the poses' hwf column."""
    imgs = np.asarray(images)
    poses, bds, sc = load_poses(poses_bounds, factor, imgs.shape[1:], bd_factor=bd_factor)
    c2w = None
    if recenter:
        poses, c2w = (recenter_poses(poses, c2w=recenter_partial), recenter_partial) if recenter_partial is not None else recenter_poses(poses, return_c2w=True)
    if K is None:
        H, W_, focal = poses[0, :3, -1]
        K = np.array([[focal, 0, 0.5 * W_], [0, focal, 0.5 * H], [0, 0, 1]])
    return ImageBatcher(imgs, poses[:, :3, :4], K, device=device), {"poses": poses, "bds": bds, "sc": sc, "recenter_partial": c2w}
