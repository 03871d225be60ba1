"""The losses of a training step that need no dataset (train.py:326-441), shared by tests/golden/make_golden.py (which evaluates them on the
reference's render_decomp output) and the GPU tests (which evaluate them on the HIP path's): radiance on both passes (:332), the three coarse
radiances (:336-341), albedo prior (:396), irradiance prior and regulariser (:399, :404), and — from iteration
N_iter_ignore_approximated_radiance on — the approximated radiance (color_map, :330), before it the roughness initialisation (:411-412).
Targets are seeded arrays (a loss is a loss); `calculate_loss` adds the coarse '0' map when the result holds one (:299-320)."""
import numpy as np

BETA = dict(radiance=1.0, render=1.0, prior_albedo=1.0, prior_irradiance=0.5, irradiance_reg=0.1, roughness=1.0, irr_mean=0.4, roughness_init=0.5, normal=0.5, depth=0.25)


def targets(rng, n_rays):
    tg = {k: rng.uniform(0.05, 0.95, (n_rays, 3)).astype(np.float32) for k in ("rgb", "rgb_1", "rgb_2", "rgb_3", "albedo")}
    tg["irradiance"] = rng.uniform(0.1, 0.9, (n_rays, 1)).astype(np.float32)
    return tg


def aux_targets(rng, n_rays):
    """the extra target of a step with auxiliary networks: unit normals"""
    v = rng.normal(size=(n_rays, 3)).astype(np.float32)
    return {"normal": (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)}


def total_loss(torch, res, tg, approximate_radiance, beta=BETA):
    mse = torch.nn.functional.mse_loss
    dev = res["radiance_map"].device

    def both(key, target):
        if isinstance(target, float):
            return sum(torch.mean((res[k] - target) ** 2) for k in (key, key + "0") if k in res)
        t = torch.as_tensor(target, device=dev)
        t = t.reshape(t.shape[0], -1)

        def one(x):      # (irradiance_map is [n, 3] under calculate_irradiance_from_gt: its [n, 1] target broadcasts)
            x = x.reshape(t.shape[0], -1)
            return mse(x, t if t.shape == x.shape else t.expand_as(x))
        return sum(one(res[k]) for k in (key, key + "0") if k in res)

    loss = beta["radiance"] * both("radiance_map", tg["rgb"])
    for k in range(3):
        loss = loss + beta["radiance"] * both("radiance_map_%d" % (k + 1), tg["rgb_%d" % (k + 1)])
    loss = loss + beta["prior_albedo"] * both("albedo_map", tg["albedo"])
    loss = loss + beta["prior_irradiance"] * both("irradiance_map", tg["irradiance"])
    loss = loss + beta["irradiance_reg"] * mse(res["irradiance_map"], torch.ones_like(res["irradiance_map"]) * beta["irr_mean"])
    if "inferred_normal_map" in res and "normal" in tg:      # infer_normal: the normal_mlp's composited output against a target normal (train.py's normal loss)
        loss = loss + beta.get("normal", 0.5) * both("inferred_normal_map", tg["normal"])
    if "inferred_depth_map" in res and getattr(res["inferred_depth_map"], "requires_grad", False):      # infer_depth (train.py:351): against the rendered depth, detached
        loss = loss + beta.get("depth", 0.25) * mse(res["inferred_depth_map"], res["depth_map"].detach())
    if approximate_radiance:
        loss = loss + beta["render"] * both("color_map", tg["rgb"])
    else:
        loss = loss + beta["roughness"] * both("roughness_map", beta["roughness_init"])
    return loss
