#!/usr/bin/env python3
"""Generates tests/golden/*.pt by running the REFERENCE'S OWN nn/nets.py + nn/net_blocks.py +
nn/metrics/composed_loss.py, imported from /root/reference in the build container.

Only runnable where /root/reference exists (never on the GPU box).  The seven third-party modules the
reference imports but this image lacks are replaced by the stand-ins in oracle/refgen/stubs/ (functional:
torch_geometric.nn.DynamicEdgeConv / global_*_pool and sparsemax.Sparsemax, which forward to
oracle/ref_path.py; inert: entmax, igl, customconfig, pattern).  Everything else that runs — MLP layout,
constructors/config logic, initialisers, LSTM decoders, output slicing, ComposedPatternLoss/PanelLoopLoss,
autograd — is reference code.  The fixtures hold DATA only: configs, seeds, inputs, outputs, gradients.

    python oracle/refgen/make_golden.py          # writes tests/golden/
"""
import copy
import os
import sys

import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = '/root/reference'
sys.path[:0] = [os.path.join(REPO, 'oracle', 'refgen', 'stubs'), os.path.join(REF, 'nn'), REPO]

import nets as ref_nets  # noqa: E402  (the reference's module)

torch.set_num_threads(1)   # fixtures are single-thread results

SMALL_NN = {
    'EConv_hidden': 32, 'EConv_feature': 24, 'k_neighbors': 4,
    'panel_encoding_size': 40, 'panel_hidden_size': 40, 'pattern_encoding_size': 40,
    'pattern_hidden_size': 40}


def load_cfg(rel):
    with open(os.path.join(REF, rel)) as f:
        cfg = yaml.safe_load(f)
    data_config = dict(cfg['dataset'])
    data_config['max_pattern_len'] = 23  # nn/data/datasets.py:377-379 with panel_classes_condenced.json
    return data_config, cfg['NN']


def run_case(model_name, yaml_rel, nn_override, B, N, seed, tag, keep_state):
    data_config, nn_cfg = load_cfg(yaml_rel)
    nn_cfg = copy.deepcopy(nn_cfg)
    nn_cfg.update(nn_override)
    loss_cfg = copy.deepcopy(nn_cfg['loss'])
    # stitch/free_class terms only switch on at epoch >= 40; the timed unit runs at epoch 0
    torch.manual_seed(seed)
    model = getattr(ref_nets, model_name)(data_config, copy.deepcopy(nn_cfg), copy.deepcopy(loss_cfg))
    model.loss.with_quality_eval = False
    model.train()
    g = torch.Generator().manual_seed(seed + 1)
    P, L = data_config['max_pattern_len'], data_config['max_panel_len']
    feats = torch.randn(B, N, 3, generator=g)
    gt = {'outlines': torch.randn(B, P, L, 4, generator=g), 'rotations': torch.randn(B, P, 4, generator=g),
          'translations': torch.randn(B, P, 3, generator=g),
          'num_edges': torch.randint(0, L + 1, (B, P), generator=g)}   # includes < 3 (skipped panels)
    state0 = copy.deepcopy(model.state_dict())
    torch.manual_seed(seed + 2)            # fixes the random LSTM h0/c0 draw
    preds = model(feats, log_step=0, epoch=0)
    loss, loss_dict, _ = model.loss(preds, {k: v.clone() for k, v in gt.items()}, epoch=0)
    loss.backward()
    knn = [c.last_knn.to(torch.int32) for c in model.feature_extractor.conv_layers]
    fx = {
        'model': model_name, 'yaml': yaml_rel, 'nn_override': nn_override, 'data_config': data_config,
        'nn_config': nn_cfg, 'loss_config': loss_cfg, 'B': B, 'N': N, 'seed': seed,
        'features': feats, 'gt': gt,
        'preds': {k: v.detach().clone() for k, v in preds.items()},
        'loss': loss.detach().clone(), 'loss_dict': {k: v.detach().clone() for k, v in loss_dict.items()},
        'knn': knn,
        'grad_norms': {n: p.grad.norm().item() for n, p in model.named_parameters() if p.grad is not None},
        'none_grads': [n for n, p in model.named_parameters() if p.grad is None],
        'state_keys': [(k, tuple(v.shape)) for k, v in state0.items()],
        'merged_config_keys': sorted(k for k in model.config.keys()),
        'bn_after': {k: v.clone() for k, v in model.state_dict().items()
                     if 'conv_layers.0.nn.0.2.running' in k or 'num_batches' in k and 'conv_layers.0.nn.0' in k},
        'torch': torch.__version__, 'threads': 1,
    }
    if keep_state:
        fx['state_dict'] = state0
        fx['grads'] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    out = os.path.join(REPO, 'tests', 'golden', tag + '.pt')
    torch.save(fx, out)
    print('%-28s loss=%.6f  %d state entries  %.1f KB' % (tag, loss.item(), len(state0),
                                                          os.path.getsize(out) / 1024))


if __name__ == '__main__':
    os.makedirs(os.path.join(REPO, 'tests', 'golden'), exist_ok=True)
    lstm_yaml, att_yaml = 'models/baseline/lstm_stitch_tags.yaml', 'models/att/att.yaml'
    run_case('GarmentFullPattern3D', lstm_yaml, SMALL_NN, 2, 64, 100, 'full3d_small', True)
    run_case('GarmentSegmentPattern3D', att_yaml, SMALL_NN, 2, 64, 200, 'segment3d_small', True)
    # shipped hyper-parameters; weights are re-derivable from the seed, so only results are stored
    run_case('GarmentFullPattern3D', lstm_yaml, {}, 2, 128, 300, 'full3d_shipped', False)
    run_case('GarmentSegmentPattern3D', att_yaml, {}, 2, 128, 400, 'segment3d_shipped', False)
    run_case('GarmentFullPattern3D', lstm_yaml, {'k_neighbors': 16}, 2, 256, 500, 'full3d_k16', False)
