"""
ctypes binding of libga_ops.so (include/ga_ops.h).  The product path has NO fallback: if the library is missing or
fails to load, importing this module raises.

Structures mirror include/ga_ops.h field for field; tests/test_abi.py checks sizeof(ga_op) against the library.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm bundles its own libamdhip64.so.7; libga_ops.so needs the same SONAME.  Importing torch FIRST makes the
# dynamic loader bind our library to the HIP runtime torch already initialised (one runtime per process: streams and
# device pointers are shared).  Loading ours first would pull the system ROCm copy and leave two runtimes in the process
# ("no ROCm-capable device is detected" on the second one).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GA_OPS_LIB') or os.path.join(_HERE, 'libga_ops.so')   # GA_OPS_LIB: debug builds only (make trace)

GA_ACT_NONE, GA_ACT_SILU, GA_ACT_ELU, GA_ACT_RELU, GA_ACT_LRELU, GA_ACT_FLRELU = 0, 1, 2, 3, 4, 5
GA_CONV_ADDEND_RELU, GA_CONV_ADDEND_PRE_DACT, GA_CONV_PRO_PRELU, GA_CONV_DACT_PRELU = 1, 2, 4, 8
(GA_OP_CONV, GA_OP_DWCONV5, GA_OP_REDUCE, GA_OP_SE_EXCITE, GA_OP_SE_APPLY, GA_OP_BILINEAR_BWD, GA_OP_SAMPLER,
 GA_OP_DML, GA_OP_MAXPOOL, GA_OP_IMAGE_IO, GA_OP_AXPBY, GA_OP_BLUR, GA_OP_REP_SUM, GA_OP_INTERLEAVE2,
 GA_OP_MAXPOOL3S2, GA_OP_AVGPOOL_ACT, GA_OP_GCONV, GA_OP_PRELU, GA_OP_UNARY, GA_OP_MODOUT, GA_OP_UP2_BLUR, GA_OP_PIXELNORM,
 GA_OP_LATENT_MIX, GA_OP_POOL_DENORM, GA_OP_ATTN, GA_OP_LAYERNORM, GA_OP_RESIZE2_CROP, GA_OP_DEC_CELL, GA_OP_AVAE, GA_OP_DEC_CELL_HALO) = range(1, 31)
GA_AVAE_ADAIN, GA_AVAE_AVGPOOL, GA_AVAE_PIXELNORM, GA_AVAE_SAMPLE = 0, 1, 2, 3
ABI_VERSION = 7     # include/ga_ops.h: GA_ABI_VERSION (descriptor layouts + entry points); _load() refuses any other library
ERRORS = {0: 'GA_OK', -1: 'GA_E_BADARG', -2: 'GA_E_ALIGN', -3: 'GA_E_UNSUPPORTED', -4: 'GA_E_LAUNCH'}

fp = C.c_void_p     # device pointers travel as integers
i32 = C.c_int
f32 = C.c_float


class ConvDesc(C.Structure):
    _fields_ = [('x', fp), ('ldx', i32), ('x2', fp), ('ldx2', i32), ('w', fp), ('bias', fp),
                ('pro_scale', fp), ('pro_shift', fp), ('addend', fp), ('ldadd', i32),
                ('addend2', fp), ('ldadd2', i32), ('dact_x', fp), ('lddact', i32),
                ('dact_scale', fp), ('dact_shift', fp), ('y', fp), ('ldy', i32),
                ('N', i32), ('Hi', i32), ('Wi', i32), ('C1', i32), ('C2', i32),
                ('Ho', i32), ('Wo', i32), ('Cout', i32),
                ('KH', i32), ('KW', i32), ('sn', i32), ('sd', i32), ('pad', i32),
                ('pro_act', i32), ('pro_per_row', i32), ('dact_act', i32), ('addend_bcast_n', i32), ('tile', i32),
                ('splits', i32), ('ws', fp), ('ws_floats', C.c_long),
                ('x_bytes', C.c_uint), ('x2_bytes', C.c_uint), ('w_bytes', C.c_uint), ('dact_rep', i32),
                ('w_hi', fp), ('w_lo', fp), ('addend_rep', i32), ('flags', i32), ('w_frag', fp)]


class DwDesc(C.Structure):
    _fields_ = [('x', fp), ('w', fp), ('bias', fp), ('dact_x', fp), ('y', fp),
                ('N', i32), ('H', i32), ('W', i32), ('C', i32),
                ('pro_act', i32), ('dact_act', i32), ('up2', i32), ('pool2', i32), ('act_rep', i32), ('_reserved', i32)]


class ReduceDesc(C.Structure):
    _fields_ = [('a', fp), ('b', fp), ('out', fp), ('N', i32), ('P', i32), ('C', i32), ('scale', f32),
                ('ws', fp), ('ws_floats', C.c_long), ('gate', fp), ('skip', fp), ('scaled', fp), ('a_src', fp), ('a_w', fp)]


class SeExciteDesc(C.Structure):
    _fields_ = [('m', fp), ('w1', fp), ('b1', fp), ('w2', fp), ('b2', fp), ('hid', fp), ('gate', fp),
                ('dgate', fp), ('pro_scale', fp), ('pro_shift', fp),
                ('N', i32), ('C', i32), ('Hd', i32), ('P', i32), ('res_scale', f32), ('backward', i32),
                ('t', fp), ('dout', fp), ('skip', fp), ('out', fp), ('act_rep', i32), ('_reserved', i32)]


class SeApplyDesc(C.Structure):
    _fields_ = [('skip', fp), ('t', fp), ('gate', fp), ('out', fp),
                ('N', i32), ('H', i32), ('W', i32), ('C', i32), ('skip_mode', i32), ('res_scale', f32)]


class BilinearBwdDesc(C.Structure):
    _fields_ = [('dhigh', fp), ('dlow', fp), ('N', i32), ('h', i32), ('w', i32), ('C', i32), ('accumulate', i32)]


class SamplerDesc(C.Structure):
    _fields_ = [('mu_q', fp), ('ldq', i32), ('p', fp), ('ldp', i32), ('eps', fp), ('eps_nchw', i32),
                ('z', fp), ('dz', fp), ('dmu_q', fp), ('dp', fp),
                ('N', i32), ('h', i32), ('w', i32), ('NL', i32),
                ('alpha', f32), ('one_minus_alpha', f32), ('temp', f32), ('backward', i32),
                ('q_rep', i32), ('dmu_q_rows', fp), ('ldz', i32), ('act_rep', i32), ('mode', i32), ('_reserved', i32)]


class DmlDesc(C.Structure):
    _fields_ = [('logits', fp), ('ld', i32), ('nmix', i32), ('img_nchw', fp), ('img_nhwc', fp),
                ('dimg_nhwc', fp), ('dimg_nchw', fp), ('dlogits', fp),
                ('N', i32), ('H', i32), ('W', i32), ('backward', i32), ('ld_img', i32), ('act_rep', i32)]


class MaxpoolDesc(C.Structure):
    _fields_ = [('x', fp), ('y', fp), ('dy', fp), ('dx', fp), ('N', i32), ('H', i32), ('W', i32), ('C', i32),
                ('backward', i32), ('act_rep', i32)]


class ImageIoDesc(C.Structure):
    _fields_ = [('x_nchw', fp), ('noise_nchw', fp), ('noise_coef', fp), ('y_nhwc', fp), ('dy_nhwc', fp),
                ('dx_nchw', fp), ('N', i32), ('C', i32), ('H', i32), ('W', i32), ('rep', i32), ('backward', i32),
                ('ld', i32), ('s2d', i32), ('cot_rep', i32), ('_reserved', i32)]


class AxpbyDesc(C.Structure):
    _fields_ = [('x', fp), ('y', fp), ('n', C.c_long), ('alpha', f32), ('beta', f32)]


class BlurDesc(C.Structure):
    _fields_ = [('x', fp), ('y', fp), ('taps', fp), ('planes', i32), ('H', i32), ('W', i32), ('k', i32), ('backward', i32),
                ('radius', i32), ('tmp', fp)]


class RepSumDesc(C.Structure):
    _fields_ = [('x', fp), ('y', fp), ('rows', C.c_long), ('inner', C.c_long), ('rep', i32), ('accumulate', i32)]


class Interleave2Desc(C.Structure):
    _fields_ = [('s', fp * 4), ('y', fp), ('dact_x', fp), ('dact_scale', fp), ('dact_shift', fp), ('addend', fp),
                ('addend2', fp), ('N', i32), ('H', i32), ('W', i32), ('C', i32), ('dact_act', i32), ('dact_prelu', i32),
                ('lds', i32), ('dact_rep', i32)]


class Maxpool3s2Desc(C.Structure):
    _fields_ = [('x', fp), ('y', fp), ('dy', fp), ('dx', fp), ('N', i32), ('H', i32), ('W', i32), ('C', i32),
                ('backward', i32)]


class AvgpoolActDesc(C.Structure):
    _fields_ = [('x', fp), ('y', fp), ('dy', fp), ('dx', fp), ('N', i32), ('P', i32), ('C', i32), ('act', i32),
                ('backward', i32), ('_reserved', i32)]


class GconvDesc(C.Structure):
    _fields_ = [('x', fp), ('w', fp), ('bias', fp), ('dact_x', fp), ('y', fp),
                ('N', i32), ('Hi', i32), ('Wi', i32), ('Ho', i32), ('Wo', i32), ('C', i32), ('cg', i32),
                ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32), ('pro_act', i32), ('dact_act', i32), ('_reserved', i32)]


class PreluDesc(C.Structure):
    _fields_ = [('x', fp), ('slope', fp), ('y', fp), ('dy', fp), ('dx', fp), ('rows', C.c_long), ('C', i32), ('backward', i32)]


class UnaryDesc(C.Structure):
    _fields_ = [('x', fp), ('g', fp), ('y', fp), ('n', C.c_long), ('mode', i32), ('eps', f32)]


class ModoutDesc(C.Structure):
    _fields_ = [('t', fp), ('scale', fp), ('add', fp), ('out', fp), ('dout', fp), ('dt', fp),
                ('N', i32), ('P', i32), ('C', i32), ('act', i32), ('backward', i32), ('W', i32), ('dt_planes', fp * 4),
                ('ld_planes', i32), ('_reserved2', i32), ('red', fp), ('ws', fp), ('ws_floats', C.c_long), ('t_planes', fp * 4)]


class Up2BlurDesc(C.Structure):
    _fields_ = [('lo_in', fp), ('hi', fp), ('hi_in', fp), ('lo', fp), ('N', i32), ('H', i32), ('W', i32), ('C', i32),
                ('backward', i32), ('_reserved', i32)]


class PixelnormDesc(C.Structure):
    _fields_ = [('x', fp), ('y', fp), ('rows', C.c_long), ('C', i32)]


class LatentMixDesc(C.Structure):
    _fields_ = [('codes', fp), ('avg', fp), ('styles', fp), ('alpha', fp), ('out', fp), ('dout', fp), ('dcodes', fp),
                ('R', i32), ('J', i32), ('D', i32), ('backward', i32), ('rep', i32), ('_reserved', i32)]


class PoolDenormDesc(C.Structure):
    _fields_ = [('x', fp), ('y', fp), ('dy', fp), ('dx', fp), ('N', i32), ('H', i32), ('W', i32), ('k', i32), ('ld', i32),
                ('backward', i32), ('dy_nchw', fp), ('band', i32), ('_reserved', i32)]


class AttnDesc(C.Structure):
    _fields_ = [('q', fp), ('k', fp), ('v', fp), ('out', fp), ('p', fp), ('dout', fp), ('ds', fp), ('dq', fp), ('dk', fp), ('dv', fp),
                ('ldq', i32), ('ldk', i32), ('ldv', i32), ('ldo', i32), ('lddq', i32), ('lddk', i32), ('lddv', i32),
                ('N', i32), ('Tq', i32), ('Tk', i32), ('heads', i32), ('dh', i32), ('scale', f32), ('backward', i32)]


class LayernormDesc(C.Structure):
    _fields_ = [('a', fp), ('b', fp), ('gamma', fp), ('beta', fp), ('y', fp), ('stats', fp), ('dy', fp), ('dx', fp),
                ('rows', C.c_long), ('C', i32), ('eps', f32), ('backward', i32), ('accumulate', i32)]


class Resize2CropDesc(C.Structure):
    _fields_ = [('x', fp), ('y', fp), ('dy', fp), ('dx', fp), ('N', i32), ('H', i32), ('W', i32), ('C', i32), ('crop', i32),
                ('backward', i32), ('accumulate', i32), ('_reserved', i32)]


class DecCellDesc(C.Structure):
    _fields_ = [('x', fp), ('w1_hi', fp), ('w1_lo', fp), ('b1', fp), ('wd', fp), ('wd_bwd', fp), ('bd', fp),
                ('w2_hi', fp), ('w2_lo', fp), ('b2', fp), ('dout', fp), ('pro_scale', fp), ('pro_shift', fp), ('y', fp),
                ('N', i32), ('H', i32), ('W', i32), ('C', i32), ('Hd', i32), ('backward', i32), ('act_rep', i32), ('variant', i32)]


class DecCellHaloDesc(C.Structure):
    _fields_ = [('x', fp), ('w1_hi', fp), ('w1_lo', fp), ('b1', fp), ('wd', fp), ('wd_bwd', fp), ('bd', fp),
                ('w2_hi', fp), ('w2_lo', fp), ('b2', fp), ('w1t_hi', fp), ('w1t_lo', fp), ('dout', fp), ('pro_scale', fp), ('pro_shift', fp),
                ('addend', fp), ('addend2', fp), ('y', fp),
                ('N', i32), ('H', i32), ('W', i32), ('Cin', i32), ('Cout', i32), ('Hd', i32), ('backward', i32), ('up', i32)]


class AvaeDesc(C.Structure):
    _fields_ = [('x', fp), ('a', fp), ('b', fp), ('c', fp), ('s', fp), ('dy', fp), ('y', fp), ('y2', fp),
                ('mode', i32), ('backward', i32), ('N', i32), ('P', i32), ('C', i32), ('k', i32), ('H', i32), ('W', i32),
                ('f0', f32), ('_reserved', i32)]


class _OpUnion(C.Union):
    _fields_ = [('conv', ConvDesc), ('dw', DwDesc), ('red', ReduceDesc), ('se', SeExciteDesc), ('app', SeApplyDesc),
                ('bil', BilinearBwdDesc), ('smp', SamplerDesc), ('dml', DmlDesc), ('mp', MaxpoolDesc),
                ('io', ImageIoDesc), ('ax', AxpbyDesc), ('blur', BlurDesc), ('rs', RepSumDesc), ('il', Interleave2Desc),
                ('mp3', Maxpool3s2Desc), ('ap', AvgpoolActDesc), ('gc', GconvDesc), ('pr', PreluDesc), ('un', UnaryDesc), ('mo', ModoutDesc), ('ub', Up2BlurDesc), ('lm', LatentMixDesc),
                ('pd', PoolDenormDesc), ('at', AttnDesc), ('ln', LayernormDesc), ('rc', Resize2CropDesc), ('dc', DecCellDesc), ('pn', PixelnormDesc), ('av', AvaeDesc), ('dh', DecCellHaloDesc)]


class Op(C.Structure):
    _fields_ = [('kind', i32), ('_pad', i32), ('u', _OpUnion)]


_KIND_FIELD = {GA_OP_CONV: 'conv', GA_OP_DWCONV5: 'dw', GA_OP_REDUCE: 'red', GA_OP_SE_EXCITE: 'se',
               GA_OP_SE_APPLY: 'app', GA_OP_BILINEAR_BWD: 'bil', GA_OP_SAMPLER: 'smp', GA_OP_DML: 'dml',
               GA_OP_MAXPOOL: 'mp', GA_OP_IMAGE_IO: 'io', GA_OP_AXPBY: 'ax', GA_OP_BLUR: 'blur', GA_OP_REP_SUM: 'rs',
               GA_OP_INTERLEAVE2: 'il', GA_OP_MAXPOOL3S2: 'mp3', GA_OP_AVGPOOL_ACT: 'ap', GA_OP_GCONV: 'gc', GA_OP_PRELU: 'pr', GA_OP_UNARY: 'un', GA_OP_MODOUT: 'mo', GA_OP_UP2_BLUR: 'ub', GA_OP_PIXELNORM: 'pn',
               GA_OP_LATENT_MIX: 'lm', GA_OP_POOL_DENORM: 'pd', GA_OP_ATTN: 'at', GA_OP_LAYERNORM: 'ln', GA_OP_RESIZE2_CROP: 'rc', GA_OP_DEC_CELL: 'dc', GA_OP_AVAE: 'av', GA_OP_DEC_CELL_HALO: 'dh'}
_DESC_KIND = {ConvDesc: GA_OP_CONV, DwDesc: GA_OP_DWCONV5, ReduceDesc: GA_OP_REDUCE, SeExciteDesc: GA_OP_SE_EXCITE,
              SeApplyDesc: GA_OP_SE_APPLY, BilinearBwdDesc: GA_OP_BILINEAR_BWD, SamplerDesc: GA_OP_SAMPLER,
              DmlDesc: GA_OP_DML, MaxpoolDesc: GA_OP_MAXPOOL, ImageIoDesc: GA_OP_IMAGE_IO, AxpbyDesc: GA_OP_AXPBY,
              BlurDesc: GA_OP_BLUR, RepSumDesc: GA_OP_REP_SUM, Interleave2Desc: GA_OP_INTERLEAVE2,
              Maxpool3s2Desc: GA_OP_MAXPOOL3S2, AvgpoolActDesc: GA_OP_AVGPOOL_ACT, GconvDesc: GA_OP_GCONV,
              PreluDesc: GA_OP_PRELU, UnaryDesc: GA_OP_UNARY, ModoutDesc: GA_OP_MODOUT, Up2BlurDesc: GA_OP_UP2_BLUR,
              PixelnormDesc: GA_OP_PIXELNORM, LatentMixDesc: GA_OP_LATENT_MIX, PoolDenormDesc: GA_OP_POOL_DENORM,
              AttnDesc: GA_OP_ATTN, LayernormDesc: GA_OP_LAYERNORM, Resize2CropDesc: GA_OP_RESIZE2_CROP, DecCellDesc: GA_OP_DEC_CELL, AvaeDesc: GA_OP_AVAE, DecCellHaloDesc: GA_OP_DEC_CELL_HALO}

EXPORTS = ['ga_conv2d', 'ga_dwconv5', 'ga_rowchan_reduce', 'ga_se_excite', 'ga_se_apply', 'ga_bilinear_up2_bwd',
           'ga_sampler_mix', 'ga_dml_mean', 'ga_maxpool2', 'ga_image_io', 'ga_axpby', 'ga_plan_run', 'ga_plan_time',
           'ga_plan_profile', 'ga_split_bf16', 'ga_gauss_blur', 'ga_rep_sum', 'ga_interleave2', 'ga_maxpool3s2', 'ga_avgpool_act', 'ga_gconv', 'ga_prelu', 'ga_unary', 'ga_modout', 'ga_up2_blur', 'ga_pixelnorm', 'ga_latent_mix', 'ga_pool_denorm', 'ga_attn', 'ga_layernorm', 'ga_resize2_crop', 'ga_dec_cell', 'ga_dec_cell_supported', 'ga_dec_cell_halo', 'ga_dec_cell_halo_supported', 'ga_avae', 'ga_microbench_hbm_copy', 'ga_microbench_mfma_bf16', 'ga_microbench_mfma_bf16_shape', 'ga_graph_capture', 'ga_graph_launch', 'ga_graph_destroy',
           'ga_last_hip_error', 'ga_abi_version', 'ga_sizeof_op', 'ga_debug_set_conv_row_limit']


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          f'or `make -C gen_adversarial_amd/csrc` — there is no CPU fallback on the product path')
    lib = C.CDLL(LIB_PATH)
    for n, d in (('ga_conv2d', ConvDesc), ('ga_dwconv5', DwDesc), ('ga_rowchan_reduce', ReduceDesc),
                 ('ga_se_excite', SeExciteDesc), ('ga_se_apply', SeApplyDesc), ('ga_bilinear_up2_bwd', BilinearBwdDesc),
                 ('ga_sampler_mix', SamplerDesc), ('ga_dml_mean', DmlDesc), ('ga_maxpool2', MaxpoolDesc),
                 ('ga_image_io', ImageIoDesc), ('ga_gauss_blur', BlurDesc), ('ga_interleave2', Interleave2Desc),
                 ('ga_maxpool3s2', Maxpool3s2Desc), ('ga_avgpool_act', AvgpoolActDesc), ('ga_gconv', GconvDesc), ('ga_prelu', PreluDesc),
                 ('ga_unary', UnaryDesc), ('ga_modout', ModoutDesc), ('ga_up2_blur', Up2BlurDesc),
                 ('ga_latent_mix', LatentMixDesc), ('ga_pool_denorm', PoolDenormDesc), ('ga_attn', AttnDesc),
                 ('ga_layernorm', LayernormDesc), ('ga_resize2_crop', Resize2CropDesc), ('ga_dec_cell', DecCellDesc), ('ga_avae', AvaeDesc), ('ga_dec_cell_halo', DecCellHaloDesc)):
        f = getattr(lib, n)
        f.argtypes = [C.POINTER(d), C.c_void_p]
        f.restype = C.c_int
    lib.ga_axpby.argtypes = [fp, fp, C.c_long, f32, f32, C.c_void_p]
    lib.ga_axpby.restype = C.c_int
    lib.ga_plan_run.argtypes = [C.POINTER(Op), C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    lib.ga_plan_run.restype = C.c_int
    lib.ga_plan_time.argtypes = [C.POINTER(Op), C.c_int, C.c_void_p, C.c_int, C.POINTER(f32), C.POINTER(f32),
                                 C.POINTER(C.c_long)]
    lib.ga_plan_time.restype = C.c_int
    lib.ga_plan_profile.argtypes = [C.POINTER(Op), C.c_int, C.c_void_p, C.POINTER(f32)]
    lib.ga_plan_profile.restype = C.c_int
    lib.ga_rep_sum.argtypes = [fp, fp, C.c_long, C.c_long, C.c_int, C.c_int, C.c_void_p]
    lib.ga_rep_sum.restype = C.c_int
    lib.ga_pixelnorm.argtypes = [fp, fp, C.c_long, C.c_int, C.c_void_p]
    lib.ga_pixelnorm.restype = C.c_int
    lib.ga_graph_capture.argtypes = [C.POINTER(Op), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.ga_graph_capture.restype = C.c_int
    lib.ga_graph_launch.argtypes = [C.c_void_p, C.c_void_p]
    lib.ga_graph_launch.restype = C.c_int
    lib.ga_graph_destroy.argtypes = [C.c_void_p]
    lib.ga_graph_destroy.restype = C.c_int
    lib.ga_split_bf16.argtypes = [fp, fp, fp, C.c_long, C.c_void_p]
    lib.ga_split_bf16.restype = C.c_int
    lib.ga_last_hip_error.restype = C.c_char_p
    lib.ga_abi_version.restype = C.c_int
    lib.ga_sizeof_op.restype = C.c_ulong
    lib.ga_debug_set_conv_row_limit.restype = C.c_long
    lib.ga_dec_cell_supported.argtypes = [C.c_int] * 5
    lib.ga_dec_cell_supported.restype = C.c_int
    lib.ga_dec_cell_halo_supported.argtypes = [C.c_int] * 5
    lib.ga_dec_cell_halo_supported.restype = C.c_int
    lib.ga_debug_set_conv_row_limit.argtypes = [C.c_long]
    lib.ga_microbench_hbm_copy.argtypes = [fp, fp, C.c_long, C.c_void_p]
    lib.ga_microbench_hbm_copy.restype = C.c_int
    lib.ga_microbench_mfma_bf16.argtypes = [fp, C.c_int, C.c_int, C.c_void_p]
    lib.ga_microbench_mfma_bf16.restype = C.c_int
    lib.ga_microbench_mfma_bf16_shape.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.ga_microbench_mfma_bf16_shape.restype = C.c_int
    if lib.ga_abi_version() != ABI_VERSION:
        raise ImportError(f'ABI mismatch: {LIB_PATH} reports GA_ABI_VERSION {lib.ga_abi_version()}, the binding is written for '
                          f'{ABI_VERSION}: rebuild with `make -C gen_adversarial_amd/csrc`')
    if lib.ga_sizeof_op() != C.sizeof(Op):
        raise ImportError(f'ABI mismatch: library ga_op is {lib.ga_sizeof_op()} bytes, binding is {C.sizeof(Op)}')
    return lib


lib = _load()


class GaError(RuntimeError):
    pass


def check(rc: int, what: str = ''):
    if rc != 0:
        extra = f' ({lib.ga_last_hip_error().decode()})' if rc == -4 else ''
        raise GaError(f'{what}: {ERRORS.get(rc, rc)}{extra}')


def make_op(desc) -> Op:
    op = Op()
    op.kind = _DESC_KIND[type(desc)]
    setattr(op.u, _KIND_FIELD[op.kind], desc)
    return op


_DIRECT = {ConvDesc: 'ga_conv2d', DwDesc: 'ga_dwconv5', ReduceDesc: 'ga_rowchan_reduce', SeExciteDesc: 'ga_se_excite',
           SeApplyDesc: 'ga_se_apply', BilinearBwdDesc: 'ga_bilinear_up2_bwd', SamplerDesc: 'ga_sampler_mix',
           DmlDesc: 'ga_dml_mean', MaxpoolDesc: 'ga_maxpool2', ImageIoDesc: 'ga_image_io', BlurDesc: 'ga_gauss_blur',
           Interleave2Desc: 'ga_interleave2', Maxpool3s2Desc: 'ga_maxpool3s2', AvgpoolActDesc: 'ga_avgpool_act',
           GconvDesc: 'ga_gconv', PreluDesc: 'ga_prelu', UnaryDesc: 'ga_unary', ModoutDesc: 'ga_modout', Up2BlurDesc: 'ga_up2_blur',
           LatentMixDesc: 'ga_latent_mix', PoolDenormDesc: 'ga_pool_denorm', AttnDesc: 'ga_attn', LayernormDesc: 'ga_layernorm',
           Resize2CropDesc: 'ga_resize2_crop', DecCellDesc: 'ga_dec_cell', AvaeDesc: 'ga_avae', DecCellHaloDesc: 'ga_dec_cell_halo'}


def run(desc, stream: int = 0):
    """Launch one op directly through its own C entry point."""
    if isinstance(desc, AxpbyDesc):
        check(lib.ga_axpby(desc.x, desc.y, desc.n, desc.alpha, desc.beta, stream), 'ga_axpby')
        return
    if isinstance(desc, PixelnormDesc):
        check(lib.ga_pixelnorm(desc.x, desc.y, desc.rows, desc.C, stream), 'ga_pixelnorm')
        return
    if isinstance(desc, RepSumDesc):
        check(lib.ga_rep_sum(desc.x, desc.y, desc.rows, desc.inner, desc.rep, desc.accumulate, stream), 'ga_rep_sum')
        return
    name = _DIRECT[type(desc)]
    check(getattr(lib, name)(C.byref(desc), stream), name)


class Plan:
    """A flat list of ops replayed by one ga_plan_run call."""

    def __init__(self):
        self.descs = []
        self.names = []
        self._arr = None

    def add(self, desc, name: str = ''):
        self.descs.append(desc)
        self.names.append(name)
        self._arr = None

    def finalize(self):
        arr = (Op * len(self.descs))()
        for i, d in enumerate(self.descs):
            arr[i] = make_op(d)
        self._arr = arr
        return self

    def __len__(self):
        return len(self.descs)

    def run(self, stream: int = 0, start: int = 0, end: int = None):
        """Replay ops[start:end) with one C call."""
        if self._arr is None:
            self.finalize()
        end = len(self.descs) if end is None else end
        if end <= start:
            return
        failed = C.c_int(-1)
        first = C.cast(C.byref(self._arr, start * C.sizeof(Op)), C.POINTER(Op))
        rc = lib.ga_plan_run(first, end - start, stream, C.byref(failed))
        if rc != 0:
            check(rc, f'plan op #{start + failed.value} ({self.names[start + failed.value]})')

    def capture(self, stream: int):
        """capture one replay into a HIP graph (stream must not be the NULL stream); returns an opaque handle"""
        if self._arr is None:
            self.finalize()
        h = C.c_void_p()
        check(lib.ga_graph_capture(self._arr, len(self.descs), stream, C.byref(h)), 'ga_graph_capture')
        return h

    @staticmethod
    def launch_graph(handle, stream: int):
        check(lib.ga_graph_launch(handle, stream), 'ga_graph_launch')

    @staticmethod
    def destroy_graph(handle):
        lib.ga_graph_destroy(handle)

    def profile(self, stream: int = 0):
        """per-op device milliseconds of one replay."""
        if self._arr is None:
            self.finalize()
        out = (f32 * len(self.descs))()
        check(lib.ga_plan_profile(self._arr, len(self.descs), stream, out), 'ga_plan_profile')
        return list(out)

    def time(self, stream: int = 0, iters: int = 1, per_conv: bool = False):
        if self._arr is None:
            self.finalize()
        total, conv, nconv = f32(0), f32(0), C.c_long(0)
        rc = lib.ga_plan_time(self._arr, len(self.descs), stream, iters, C.byref(total),
                              C.byref(conv) if per_conv else None, C.byref(nconv) if per_conv else None)
        check(rc, 'ga_plan_time')
        return total.value, conv.value, nconv.value
