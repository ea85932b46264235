"""OpenGL sharing entry points (cpm_gl_*: the consumer side of the light volume) on a box without a
display: the calling thread has no current OpenGL context, so every registration must be refused with CPM_ERR_UNSUPPORTED
before the HIP runtime is asked, and argument errors must be reported as such.  The device half of the hand-over
(cpm_light_volume_texels: what is written into the mapped buffer) is checked here; the mapped path itself (register ->
acquire -> copy -> release against a live context) needs the host application's context and has never run in this
repository's environment -- INTEGRATION.md says so."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CPM_ERR_INVALID_ARGUMENT, CPM_ERR_UNSUPPORTED = -1, -4


def test_no_context_is_refused_not_crashed(ctx, cpm):
    B = cpm.binding
    assert ctx.gl_available() is False
    for read_only in (False, True):
        with pytest.raises(B.CpmError, match="no current OpenGL context") as e:
            ctx.gl_register_buffer(7, read_only)
        assert e.value.status == CPM_ERR_UNSUPPORTED


def test_argument_errors(ctx, cpm):
    B = cpm.binding
    with pytest.raises(B.CpmError) as e:
        ctx.gl_register_buffer(0)
    assert e.value.status == CPM_ERR_INVALID_ARGUMENT
    ctx.gl_acquire([])   # nothing to map: not an error
    ctx.gl_release([])
    null = B.GLResource(ctx, None)
    with pytest.raises(B.CpmError) as e:
        ctx.gl_acquire([null])
    assert e.value.status == CPM_ERR_INVALID_ARGUMENT
    with pytest.raises(B.CpmError) as e:
        ctx.gl_copy_to_buffer(ctx.torch.zeros(8, device=ctx.device), null)
    assert e.value.status == CPM_ERR_INVALID_ARGUMENT
    with pytest.raises(B.CpmError) as e:
        null.pointer()
    assert e.value.status == CPM_ERR_INVALID_ARGUMENT
    v = ctx.torch.zeros(8, device=ctx.device)
    with pytest.raises(B.CpmError) as e:
        ctx.light_volume_texels(v, v, texel=9)
    assert e.value.status == CPM_ERR_INVALID_ARGUMENT
    ctx.lib.cpm_gl_unregister(ctx.h, None)  # a null resource is ignored


@pytest.mark.parametrize("n", [1, 3, 4, 1021, 128 ** 3, 4 * 33 * 7 * 5])
def test_light_volume_texels(ctx, cpm, n):
    """The device half of the hand-over (what cpm_gl_copy_to_buffer writes into the mapped pixel-unpack buffer): float32
    texels bit for bit, float16 texels = round-to-nearest-even of the light volume (numpy's conversion), subnormals and
    lengths off the vector width included."""
    B = cpm.binding
    torch = ctx.torch
    g = torch.Generator(device="cpu").manual_seed(n)
    vol = torch.rand(n, generator=g) * 3.0 - 0.5
    vol[::17] = 0.0
    vol[5::97] *= 1e-6   # float16 subnormals
    vol[7::1013] *= 1e5  # beyond float16's range: inf
    dev = vol.to(ctx.device)
    out32 = torch.empty(n, dtype=torch.float32, device=ctx.device)
    ctx.light_volume_texels(dev, out32, B.CPM_GL_TEXEL_F32)
    out16 = torch.empty(n, dtype=torch.float16, device=ctx.device)
    ctx.light_volume_texels(dev, out16, B.CPM_GL_TEXEL_F16)
    ctx.light_volume_texels(dev, dev, B.CPM_GL_TEXEL_F32)  # in place: nothing to do
    torch.cuda.synchronize()
    assert np.array_equal(out32.cpu().numpy().view(np.uint32), vol.numpy().view(np.uint32))
    assert np.array_equal(dev.cpu().numpy().view(np.uint32), vol.numpy().view(np.uint32))
    with np.errstate(over="ignore"):
        want16 = vol.numpy().astype(np.float16)
    assert np.array_equal(out16.cpu().numpy().view(np.uint16), want16.view(np.uint16))
