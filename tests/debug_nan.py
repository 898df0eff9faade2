import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gen_adversarial_amd.engine import Engine
from gen_adversarial_amd.nvae_spec import init_nvae_state_dict, build_spec
from gen_adversarial_amd.vgg_spec import build_vgg_spec, init_vgg_state_dict
CFG = {'initial_channels': 8, 'num_pre-post_process_blocks': 1, 'num_pre-post_process_cells': 2, 'num_scales': 3,
       'num_groups_per_scale': 2, 'is_adaptive': False, 'min_groups_per_scale': 1, 'num_cells_per_group': 1,
       'num_latent_per_group': 4, 'num_logistic_mixtures': 10, 'num_nf_cells': None}
RES = (3, 64, 64)
sd = init_nvae_state_dict(CFG, RES, 5); vs = build_vgg_spec(100, 16); vsd = init_vgg_state_dict(100, 16, 6)
n = len(build_spec(CFG, RES).groups)
al = [0.7 * i / (n - 1) for i in range(n)]
for prec in ('fp32', 'bf16x3'):
    eng = Engine(sd, CFG, RES, vsd, vs, rows=4, rep=4, alphas=al, noise_eps=2.0, device='cuda:0', precision=prec)
    g = torch.Generator(device='cuda').manual_seed(0)
    eng.x_in.copy_(torch.rand(1, 3, 64, 64, device='cuda', generator=g))
    for e in eng.eps: e.normal_(generator=g)
    eng.noise.normal_(generator=g); eng.noise_coef.copy_(2.0 / eng.noise.flatten(1).norm(dim=1))
    eng.forward()
    print(prec, 'logits finite', torch.isfinite(eng.logits).all().item())
    eng.dlogits.zero_(); eng.dlogits.view_as(eng.logits)[:, 0] = 0.25
    eng.backward(); torch.cuda.synchronize()
    print(prec, 'dx finite', torch.isfinite(eng.dx).all().item())
    bad = [k for k, a in eng.acts.items() if a._g is not None and not torch.isfinite(a._g).all()]
    print('non-finite grads:', bad[-6:], len(bad))
    names = eng.bwd.names
    # find the first op after which something is non finite, by replaying op by op
    eng.forward()
    for a in eng.acts.values():
        if a._g is not None and a._g.data_ptr() != eng.dlogits.data_ptr(): a._g.zero_()
    for v in eng._scratch.values(): v.zero_()
    for i in range(len(eng.bwd)):
        eng.bwd.run(eng.stream(), start=i, end=i + 1)
        torch.cuda.synchronize()
        bad = [k for k, a in eng.acts.items() if a._g is not None and not torch.isfinite(a._g).all()]
        bad += [str(k) for k, v in eng._scratch.items() if not torch.isfinite(v).all()]
        if bad:
            d = eng.bwd.descs[i]
            print('first bad after op', i, names[i], bad[:3])
            print({f[0]: getattr(d, f[0]) for f in d._fields_ if f[1].__name__ in ('c_int',)})
            break
