"""
Checkpoint ingestion with the reference's file layouts (src/defenses/loading_utils.py:10-81), returning plain weight
holders instead of nn.Modules: the arithmetic lives in libga_ops, not in torch modules.

  load_Vgg11(path, device, n_classes=100)   ckpt['state_dict'] with keys model.features.N.*, model.classifier.{0,1,3}.*
  load_ResNet50(path, device, n_classes=2)  ckpt['state_dict'] with torchvision resnet50 keys under `model.` and the
                                            projector head model.fc.{0,1,3}.* (loading_utils.py:10-16)
  load_NVAE(path, device, temperature)      ckpt['configuration'] {'autoencoder': cfg, 'resolution': (C,H,W)} and
                                            ckpt[f'state_dict_temp={temperature}']
E4E / Style-Transformer loaders are the "next" rows of SURVEY.md §8 and raise NotImplementedError.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from ..nvae_spec import build_spec
from ..resnet_spec import RESNET50_BLOCKS, build_resnet_spec
from ..vgg_spec import build_vgg_spec


@dataclass
class VggWeights:
    state_dict: Dict[str, torch.Tensor]
    n_classes: int
    width_div: int = 1

    @property
    def spec(self):
        return build_vgg_spec(self.n_classes, self.width_div)

    def to(self, device):          # nn.Module-like no-ops so that caller code written for the reference keeps working
        return self

    def eval(self):
        return self


@dataclass
class ResNetWeights:
    state_dict: Dict[str, torch.Tensor]
    n_classes: int
    width_div: int = 1
    blocks: Tuple[int, ...] = RESNET50_BLOCKS
    groups: int = 1
    width_per_group: int = 64

    @property
    def spec(self):
        return build_resnet_spec(self.n_classes, self.width_div, self.blocks, self.groups, self.width_per_group)

    def to(self, device):
        return self

    def eval(self):
        return self


@dataclass
class NVAEWeights:
    state_dict: Dict[str, torch.Tensor]
    config: dict
    resolution: Tuple[int, int, int]

    @property
    def spec(self):
        return build_spec(self.config, self.resolution)

    @property
    def num_scales(self):
        return self.config['num_scales']

    def to(self, device):
        return self

    def eval(self):
        return self


def _torch_load(path):
    return torch.load(path, map_location='cpu', weights_only=False)


def load_Vgg11(path: str, device: str, n_classes: int = 100) -> VggWeights:
    ckpt = _torch_load(path)
    sd = ckpt['state_dict']
    width_div = 64 // sd['model.features.0.weight'].shape[0]
    # the head in the checkpoint decides the class count (the reference builds Vgg(n_classes=100) and load_state_dict would refuse any
    # other head; a checkpoint with another head is a test / user model, not an error here)
    return VggWeights(sd, sd['model.classifier.3.weight'].shape[0], width_div)


def load_NVAE(checkpoint_path: str, device: str, temperature: float) -> NVAEWeights:
    ckpt = _torch_load(checkpoint_path)
    config = ckpt['configuration']
    return NVAEWeights(ckpt[f'state_dict_temp={temperature}'], dict(config['autoencoder']), tuple(config['resolution']))


def load_ResNet50(path: str, device: str, n_classes: int = 2) -> ResNetWeights:
    """src/defenses/loading_utils.py:10-16.  Width and depth are read off the tensors, so reduced test checkpoints load too."""
    sd = _torch_load(path)['state_dict']
    width_div = 64 // sd['model.conv1.weight'].shape[0]
    blocks = tuple(1 + max(int(k.split('.')[2]) for k in sd if k.startswith(f'model.layer{i}.')) for i in (1, 2, 3, 4))
    return ResNetWeights(sd, sd['model.fc.3.weight'].shape[0] if n_classes is None else n_classes, width_div, blocks)


def load_ResNext50(path: str, device: str, n_classes: int = 4) -> ResNetWeights:
    """src/defenses/loading_utils.py:28-34: resnext50_32x4d (groups 32, 4 channels per group at the first stage)."""
    sd = _torch_load(path)['state_dict']
    width_div = 64 // sd['model.conv1.weight'].shape[0]
    blocks = tuple(1 + max(int(k.split('.')[2]) for k in sd if k.startswith(f'model.layer{i}.')) for i in (1, 2, 3, 4))
    w2 = sd['model.layer1.0.conv2.weight']
    groups = w2.shape[0] // w2.shape[1]
    return ResNetWeights(sd, sd['model.fc.3.weight'].shape[0] if n_classes is None else n_classes, width_div, blocks, groups,
                         width_per_group=w2.shape[1] * width_div)


@dataclass
class E4EWeights:
    """what pSp holds for the defender (StyleGan_E4E/psp.py:15-125): encoder / decoder state dicts (prefixes stripped),
    latent_avg, the training options; specs are derived from the tensor shapes"""
    encoder_sd: Dict[str, torch.Tensor]
    encoder_spec: object
    decoder_sd: Dict[str, torch.Tensor]
    decoder_spec: object
    latent_avg: Optional[torch.Tensor]
    opts: dict

    def to(self, device):
        return self

    def eval(self):
        return self


def estimate_latent_avg(decoder_sd: Dict[str, torch.Tensor], gspec, device: str, n_latent: int = 10000, seed: Optional[int] = None) -> torch.Tensor:
    """pSp.__load_latent_avg for a checkpoint without 'latent_avg' (psp.py:117-125 -> Generator.mean_latent,
    stylegan2/generator.py:388-395): the mean of the mapping network over `n_latent` N(0,1) latents, [1, style_dim].  The
    mapping network runs on the HIP engine (PixelNorm + 8 FC layers, engine_stylegan.build_mapping); the draw is torch's device
    RNG (the reference's estimate is equally a random draw: the two agree statistically, not bit for bit)."""
    from ..engine import Engine
    eng = Engine.bare(n_latent, device=device, need_backward=False)
    z = eng.alloc((n_latent, gspec.style_dim))
    out = eng.build_mapping(decoder_sd, z)
    eng.finish()
    g = None
    if seed is not None:
        g = torch.Generator(device=device).manual_seed(seed)
    z.normal_(generator=g)
    eng.forward()
    return out.view(n_latent, gspec.style_dim).double().mean(dim=0, keepdim=True).float()


def load_E4EStyleGan(checkpoint_path: str, device: str) -> E4EWeights:
    """src/defenses/loading_utils.py:37-48 + pSp.load_weights (psp.py:39-46,119-125): checkpoint keys 'state_dict' (with
    'encoder.' / 'decoder.' prefixes), 'latent_avg', 'opts'.  Widths and depths are read off the tensors, so reduced test
    checkpoints load too."""
    from ..e4e_spec import build_e4e_spec
    from ..stylegan_spec import build_stylegan_spec
    ckpt = _torch_load(checkpoint_path)
    opts = dict(ckpt['opts'])
    sd = ckpt['state_dict']
    enc = {k[len('encoder.'):]: v for k, v in sd.items() if k.startswith('encoder.')}
    dec = {k[len('decoder.'):]: v for k, v in sd.items() if k.startswith('decoder.')}
    if opts.get('encoder_type', 'Encoder4Editing') != 'Encoder4Editing':
        raise NotImplementedError(f"encoder_type {opts.get('encoder_type')!r}: pSp only builds Encoder4Editing (psp.py:32-38)")
    size = int(opts['stylegan_size'])
    depths = []
    i = 0
    while f'body.{i}.res_layer.1.weight' in enc:
        depths.append(enc[f'body.{i}.res_layer.1.weight'].shape[0])
        i += 1
    units = tuple(depths.count(d) for d in sorted(set(depths)))
    espec = build_e4e_spec(size, 64 // enc['input_layer.0.weight'].shape[0], units)
    c4 = dec['conv1.conv.weight'].shape[1]
    gspec = build_stylegan_spec(size, 2, 512 // c4, dec['conv1.conv.modulation.weight'].shape[1])
    for sp in (gspec.conv1, gspec.to_rgb1) + gspec.convs + gspec.to_rgbs:
        got = tuple(dec[f'{sp.prefix}.conv.weight'].shape)
        if got != (1, sp.cout, sp.cin, sp.kernel, sp.kernel):
            raise ValueError(f'decoder.{sp.prefix}.conv.weight has shape {got}, expected {(1, sp.cout, sp.cin, sp.kernel, sp.kernel)}')
    if 'latent_avg' in ckpt:
        avg = ckpt['latent_avg'].float()
        if avg.dim() == 1:
            avg = avg.view(1, -1).expand(gspec.n_latent, -1)
        avg = avg.reshape(gspec.n_latent, gspec.style_dim).contiguous()
    elif opts.get('start_from_latent_avg', False):
        avg = estimate_latent_avg(dec, gspec, device).cpu().expand(gspec.n_latent, -1).contiguous()
    else:
        avg = None
    if not opts.get('start_from_latent_avg', False):
        avg = None
    return E4EWeights(enc, espec, dec, gspec, avg, opts)


@dataclass
class TransWeights:
    """what StyleTransformer holds for the defender (StyleGan_Trans/models/style_transformer.py:16-92): encoder / decoder state
    dicts (the 'encoder.module.' / 'decoder.module.' prefixes stripped), latent_avg, the training options"""
    encoder_sd: Dict[str, torch.Tensor]
    encoder_spec: object
    decoder_sd: Dict[str, torch.Tensor]
    decoder_spec: object
    latent_avg: Optional[torch.Tensor]
    opts: dict

    def to(self, device):
        return self

    def eval(self):
        return self


def load_TranStyleGan(checkpoint_path: str, device: str) -> TransWeights:
    """src/defenses/loading_utils.py:69-81 + StyleTransformer.load_weights (style_transformer.py:30-37,84-91): checkpoint keys
    'state_dict' ('encoder.module.*', 'decoder.module.*'), 'latent_avg', 'opts' (output_size, start_from_latent_avg, learn_in_w).
    Widths and depths are read off the tensors, so reduced test checkpoints load too."""
    from ..stylegan_spec import build_stylegan_spec
    from ..trans_spec import build_trans_spec
    ckpt = _torch_load(checkpoint_path)
    opts = dict(ckpt['opts'])
    sd = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
    enc = {k[len('encoder.module.'):]: v for k, v in sd.items() if k.startswith('encoder.module.')}
    dec = {k[len('decoder.module.'):]: v for k, v in sd.items() if k.startswith('decoder.module.')}
    if not enc or not dec:
        raise KeyError("a Style-Transformer checkpoint holds 'encoder.module.*' and 'decoder.module.*' (style_transformer.py:34-35)")
    if opts.get('learn_in_w', False):
        raise NotImplementedError('learn_in_w checkpoints (one shared latent) are not what the cars defender loads')
    size = int(opts['output_size'])
    depths = []
    i = 0
    while f'body.{i}.res_layer.1.weight' in enc:
        depths.append(enc[f'body.{i}.res_layer.1.weight'].shape[0])
        i += 1
    units = tuple(depths.count(d) for d in sorted(set(depths)))
    tspec = build_trans_spec(64 // enc['input_layer.0.weight'].shape[0], units)
    if enc['z'].shape[1:] != (tspec.n_query, tspec.d_model):
        raise ValueError(f"encoder.z has shape {tuple(enc['z'].shape)}, expected (1, {tspec.n_query}, {tspec.d_model})")
    c4 = dec['conv1.conv.weight'].shape[1]
    gspec = build_stylegan_spec(size, 2, 512 // c4, dec['conv1.conv.modulation.weight'].shape[1])
    avg = None
    if opts.get('start_from_latent_avg', False) and 'latent_avg' in ckpt:
        avg = ckpt['latent_avg'].float()
        avg = avg.view(1, -1).expand(tspec.n_query, -1) if avg.dim() == 1 else avg
        avg = avg.reshape(-1, tspec.d_model)[:tspec.n_query].contiguous()
    return TransWeights(enc, tspec, dec, gspec, avg, opts)
