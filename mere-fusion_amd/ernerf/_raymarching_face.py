"""`import _raymarching_face as _backend` (ernerf/raymarching/raymarching.py:10): the inference entry points."""
from . import backend as B


def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    """raymarching.py:44 -> raymarching.cu:147-155."""
    B.call("mf_near_far_from_aabb", B.f32(rays_o, "rays_o"), B.f32(rays_d, "rays_d"), B.f32(aabb, "aabb"), int(N), float(min_near),
           B.f32(nears, "nears"), B.f32(fars, "fars"), B.stream())


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, near, far, xyzs, dirs,
               deltas, noises):
    """raymarching.py:393 -> raymarching.cu:932-940."""
    B.call("mf_march_rays", int(n_alive), int(n_step), B.i32(rays_alive, "rays_alive"), B.f32(rays_t, "rays_t"), B.f32(rays_o, "rays_o"),
           B.f32(rays_d, "rays_d"), float(bound), float(dt_gamma), int(max_steps), int(C), int(H), B.u8(grid, "density_bitfield"),
           B.f32(near, "nears"), B.f32(far, "fars"), B.f32(xyzs, "xyzs"), B.f32(dirs, "dirs"), B.f32(deltas, "deltas"),
           B.f32(noises, "noises"), B.stream())


def composite_rays_triplane(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, ambs_aud, ambs_eye, uncertainties,
                            weights_sum, depth, image, amb_aud_sum, amb_eye_sum, uncertainty_sum):
    """raymarching.py:666 -> raymarching.cu:2252-2258."""
    B.call("mf_composite_rays_triplane", int(n_alive), int(n_step), float(T_thresh), B.i32(rays_alive, "rays_alive"), B.f32(rays_t, "rays_t"),
           B.f32(sigmas, "sigmas"), B.f32(rgbs, "rgbs"), B.f32(deltas, "deltas"), B.f32(ambs_aud, "ambs_aud"), B.f32(ambs_eye, "ambs_eye"),
           B.f32(uncertainties, "uncertainties"), B.f32(weights_sum, "weights_sum"), B.f32(depth, "depth"), B.f32(image, "image"),
           B.f32(amb_aud_sum, "amb_aud_sum"), B.f32(amb_eye_sum, "amb_eye_sum"), B.f32(uncertainty_sum, "uncertainty_sum"), B.stream())


def _training_only(name):
    def f(*a, **k):
        raise RuntimeError(f"_raymarching_face.{name}: training / occupancy-grid maintenance is outside the MI355X inference path")
    f.__name__ = name
    return f


for _n in ("sph_from_ray", "morton3D", "morton3D_invert", "packbits", "morton3D_dilation", "march_rays_train", "march_rays_train_backward",
           "composite_rays_train_forward", "composite_rays_train_backward", "composite_rays", "composite_rays_ambient",
           "composite_rays_train_sigma_forward", "composite_rays_train_sigma_backward", "composite_rays_ambient_sigma",
           "composite_rays_train_uncertainty_forward", "composite_rays_train_uncertainty_backward", "composite_rays_uncertainty",
           "composite_rays_train_triplane_forward", "composite_rays_train_triplane_backward"):
    globals()[_n] = _training_only(_n)
