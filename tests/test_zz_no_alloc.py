"""GPU test of pt_trace_rays_device_noalloc, the entry point that allocates nothing (include/pt_api.h): its hit
records must equal pt_trace_rays_device's bit for bit, on two calls in a row, the second with a smaller input (the
static scratch keeps the first call's counters and stack spill)."""
import numpy as np
import pytest

from pathtracer_amd import api, scenes
from tests.common import random_rays

pytestmark = pytest.mark.gpu


def test_noalloc_trace_matches_allocating_entry_point(ctx):
    import torch
    dev = torch.device("cuda", 0)
    sc = scenes.atrium(1)
    ctx.scene_create(sc)
    info = ctx.scene_info()

    def both(rays):
        n = len(rays)
        d_rays = torch.from_numpy(rays.view(np.float32).reshape(n, 8).copy()).to(dev)
        d_a = torch.full((n, 4), -7.0, dtype=torch.float32, device=dev)
        d_b = torch.full((n, 4), -9.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()                       # the fills ran on torch's stream, the library uses its own
        ctx.trace_rays_device(d_rays.data_ptr(), n, d_a.data_ptr())
        ctx.synchronize()
        ctx.trace_rays_device_noalloc(d_rays.data_ptr(), n, d_b.data_ptr())
        ctx.synchronize()
        a = d_a.cpu().numpy().view(np.uint32)
        b = d_b.cpu().numpy().view(np.uint32)
        return a, b

    big = random_rays(71, 300000, list(info.bounds_min), list(info.bounds_max), tmin=0.001, tmax=10000.0)
    a, b = both(big)
    assert (a[:, 3] != api.PT_MISS).mean() > 0.3
    assert np.array_equal(a, b)
    small = random_rays(72, 777, list(info.bounds_min), list(info.bounds_max), tmin=0.001, tmax=10000.0)
    a, b = both(small)                                  # second call in a row, smaller input
    assert np.array_equal(a, b)
    # "allocates nothing" measured, not assumed: after the warm calls above, the driver's free-memory figure must not move
    # across further noalloc calls (a private segment, had the kernel one, would be allocated by the runtime at launch;
    # the entry point also refuses to run then: extend_ns_private_bytes in kernels.hip)
    n = len(big)
    d_rays = torch.from_numpy(big.view(np.float32).reshape(n, 8).copy()).to(dev)
    d_hits = torch.empty((n, 4), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.trace_rays_device_noalloc(d_rays.data_ptr(), n, d_hits.data_ptr())
    ctx.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    for _ in range(2):
        ctx.trace_rays_device_noalloc(d_rays.data_ptr(), n, d_hits.data_ptr())
        ctx.synchronize()
        assert torch.cuda.mem_get_info(dev)[0] == free0
    del d_rays, d_hits
    # a deep hierarchy exercises the spill region that lives in the static scratch
    from tests.test_gpu_parity import deep_hierarchy_scene
    deep, rays = deep_hierarchy_scene()
    from tests.common import object_splits_only
    with object_splits_only():
        ctx.scene_create(deep)
    assert ctx.scene_info().bvh_max_depth >= 24
    a, b = both(rays)
    assert (a[:, 3] != api.PT_MISS).mean() > 0.5
    assert np.array_equal(a, b)
    ctx.trace_rays_device_noalloc(0, 0, 0)              # zero rays: a no-op
    with pytest.raises(api.PtError) as e:
        ctx.trace_rays_device_noalloc(1 << 20, 1 << 31, 1 << 20)
    assert e.value.code == api.PT_ERR_INVALID_ARG
