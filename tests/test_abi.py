"""CPU: the C-ABI library loads, exports every symbol include/ga_ops.h declares, and its descriptors reject bad
arguments before touching the GPU (no compute calls here)."""
import ctypes as C
import os
import re

from gen_adversarial_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'ga_ops.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ga_[a-z0-9_]+)\s*\(', src)))


def test_every_declared_symbol_is_exported():
    names = _declared()
    assert len(names) >= 16 and set(names) == set(L.EXPORTS)
    for n in names:
        assert getattr(L.lib, n) is not None


def test_every_direct_entry_point_takes_a_pointer_sized_stream():
    """L.run(desc, stream) passes the hipStream_t as the second argument: without argtypes ctypes would truncate it to a
    32-bit int (only the NULL stream would survive)"""
    for desc_type, name in L._DIRECT.items():
        fn = getattr(L.lib, name)
        assert fn.argtypes is not None and len(fn.argtypes) == 2, name
        assert fn.argtypes[0] is C.POINTER(desc_type) and fn.argtypes[1] is C.c_void_p, name
        assert fn.restype is C.c_int, name


def test_struct_sizes_and_version():
    hdr = open(os.path.join(ROOT, 'include', 'ga_ops.h')).read()
    declared = int(re.search(r'#define\s+GA_ABI_VERSION\s+(\d+)', hdr).group(1))
    assert L.lib.ga_abi_version() == declared == L.ABI_VERSION
    assert L.lib.ga_sizeof_op() == C.sizeof(L.Op)


def test_bad_descriptors_are_rejected_without_a_gpu():
    for desc, fn in ((L.ConvDesc(), L.lib.ga_conv2d), (L.DwDesc(), L.lib.ga_dwconv5), (L.ReduceDesc(), L.lib.ga_rowchan_reduce),
                     (L.SeExciteDesc(), L.lib.ga_se_excite), (L.SeApplyDesc(), L.lib.ga_se_apply),
                     (L.BilinearBwdDesc(), L.lib.ga_bilinear_up2_bwd), (L.SamplerDesc(), L.lib.ga_sampler_mix),
                     (L.DmlDesc(), L.lib.ga_dml_mean), (L.MaxpoolDesc(), L.lib.ga_maxpool2), (L.ImageIoDesc(), L.lib.ga_image_io),
                     (L.BlurDesc(), L.lib.ga_gauss_blur), (L.Interleave2Desc(), L.lib.ga_interleave2),
                     (L.Maxpool3s2Desc(), L.lib.ga_maxpool3s2), (L.AvgpoolActDesc(), L.lib.ga_avgpool_act),
                     (L.GconvDesc(), L.lib.ga_gconv), (L.PreluDesc(), L.lib.ga_prelu), (L.UnaryDesc(), L.lib.ga_unary),
                     (L.ModoutDesc(), L.lib.ga_modout), (L.Up2BlurDesc(), L.lib.ga_up2_blur),
                     (L.LatentMixDesc(), L.lib.ga_latent_mix), (L.PoolDenormDesc(), L.lib.ga_pool_denorm),
                     (L.AttnDesc(), L.lib.ga_attn), (L.LayernormDesc(), L.lib.ga_layernorm),
                     (L.Resize2CropDesc(), L.lib.ga_resize2_crop), (L.DecCellDesc(), L.lib.ga_dec_cell)):
        assert fn(C.byref(desc), None) == -1
    assert L.lib.ga_plan_run(None, 0, None, None) == -1
    assert L.lib.ga_pixelnorm(None, None, 4, 512, None) == -1
    m = L.ModoutDesc()                            # StyleGAN2 tail: channel count must be a multiple of 4 lanes
    m.t = m.out = 16
    m.N, m.P, m.C = 1, 4, 6
    assert L.lib.ga_modout(C.byref(m), None) == -3
    m.C, m.backward, m.dout, m.dt = 8, 1, 16, 16    # de-interleaved cotangent planes: all four or none, even H and W
    m.dt_planes[0] = 16
    assert L.lib.ga_modout(C.byref(m), None) == -1
    p = L.PoolDenormDesc()
    p.x = p.y = 16
    p.N, p.H, p.W, p.k, p.ld = 1, 3, 4, 2, 8      # odd height cannot be written in space-to-depth form
    assert L.lib.ga_pool_denorm(C.byref(p), None) == -3
    d = L.DwDesc()
    d.x = d.w = d.y = 16
    d.N = d.H = d.W = 1
    d.C = 6                                       # channel count the kernel does not implement
    assert L.lib.ga_dwconv5(C.byref(d), None) == -3
