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


ONLY = set(sys.argv[1:])      # `make_golden.py tag [tag ...]` regenerates just those fixtures


def run_case(model_name, yaml_rel, nn_override, B, N, seed, tag, keep_state, loss_override=None, gt_extra=False, epoch=0,
             stitch_gt=False):
    if ONLY and tag not in ONLY:
        return
    data_config, nn_cfg = load_cfg(yaml_rel)
    nn_cfg = copy.deepcopy(nn_cfg)
    nn_cfg.update(nn_override)
    loss_cfg = copy.deepcopy(nn_cfg['loss'])
    if loss_override:
        loss_cfg.update(loss_override)
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
    if gt_extra:   # keys the order matching permutes (nn/metrics/composed_loss.py:497-499)
        gt['empty_panels_mask'] = gt['num_edges'] < 3
    if stitch_gt:
        # what the dataset hands over for the stitch terms (nn/data/datasets.py:805-819, pattern_converter.py:81-91):
        # `stitches` [B, 2, max_num_stitches] pattern-level edge ids (panel * max_panel_len + edge), zero-padded;
        # `num_stitches` [B]; `free_edges_mask` [B, P, L] bool = edges no stitch refers to; `stitch_tags` [B, P, L, 3]
        S = data_config['max_num_stitches']
        st = torch.zeros(B, 2, S, dtype=torch.long)
        nst = torch.zeros(B, dtype=torch.long)
        free = torch.ones(B, P, L, dtype=torch.bool)
        for b in range(B):
            edges = [(p, e) for p in range(P) for e in range(int(gt['num_edges'][b, p]))]
            order = torch.randperm(len(edges), generator=g).tolist()
            n = min(int(torch.randint(3, S + 1, (1,), generator=g)), len(edges) // 2)
            nst[b] = n
            for i in range(n):
                for side in (0, 1):
                    p_, e_ = edges[order[2 * i + side]]
                    st[b, side, i] = p_ * L + e_
                    free[b, p_, e_] = False
        gt['stitches'], gt['num_stitches'], gt['free_edges_mask'] = st, nst, free
        gt['stitch_tags'] = torch.randn(B, P, L, 3, generator=g)
    if 'segmentation' in loss_cfg['loss_components']:
        # per-point panel labels (nn/data/datasets.py: `segmentation` [B, N] class ids in [0, max_pattern_len))
        gt['segmentation'] = torch.randint(0, P, (B, N), generator=g)
    state0 = copy.deepcopy(model.state_dict())
    torch.manual_seed(seed + 2)            # fixes the random LSTM h0/c0 draw
    preds = model(feats, log_step=0, epoch=epoch)
    loss, loss_dict, _ = model.loss(preds, {k: v.clone() for k, v in gt.items()}, epoch=epoch)
    loss.backward()
    knn = [c.last_knn.to(torch.int32) for c in model.feature_extractor.conv_layers]
    fx = {
        'model': model_name, 'yaml': yaml_rel, 'nn_override': nn_override, 'data_config': data_config,
        'nn_config': nn_cfg, 'loss_config': loss_cfg, 'B': B, 'N': N, 'seed': seed, 'epoch': epoch,
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
    if loss_cfg.get('panel_order_inariant_loss') or loss_cfg.get('panel_origin_invariant_loss'):
        # what the matching decided, re-derived with the reference's own helpers on the same predictions
        with torch.no_grad():
            L_ = model.loss
            L_.epoch = epoch
            L_.device = feats.device
            gt2 = {k: v.clone() for k, v in gt.items()}
            if loss_cfg.get('panel_order_inariant_loss'):
                gt2 = L_._gt_order_match(preds, gt2)
            ne = gt2['num_edges'].int().view(-1)
            fx['gt_num_edges_after_order'] = ne.clone()
            if loss_cfg.get('panel_origin_invariant_loss'):
                rot, lead = L_._batch_edge_order_match(preds['outlines'], gt2['outlines'], ne)
                fx['gt_outlines_matched'] = rot.clone()
                fx['leading_edges'] = torch.tensor([int(v) for v in lead], dtype=torch.int32)
                if stitch_gt and epoch >= loss_cfg['epoch_with_stitches']:
                    gt2 = L_._rotate_gt(preds, gt2, ne, epoch)
            else:
                fx['gt_outlines_matched'] = gt2['outlines'].clone()
            if stitch_gt and epoch >= loss_cfg['epoch_with_stitches']:
                # the re-numbered stitches / shifted free-edge mask the stitch terms were evaluated on
                fx['gt_stitches_matched'] = gt2['stitches'].clone()
                fx['gt_free_mask_matched'] = gt2['free_edges_mask'].clone()
    if keep_state:
        fx['state_dict'] = state0
        fx['grads'] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    out = os.path.join(REPO, 'tests', 'golden', tag + '.pt')
    torch.save(fx, out)
    print('%-28s loss=%.6f  %d state entries  %.1f KB' % (tag, loss.item(), len(state0),
                                                          os.path.getsize(out) / 1024))


def run_pointnet_case():
    """PointNetPlusPlus (nn/net_blocks.py:50-88) as the reference's own class, with fps / radius / PointConv from the stubs
    (restated arithmetic WITH PyG's conventions: fps(random_start=True) drawing its start points from torch's generator,
    PointConv(add_self_loops=True) re-indexing the bipartite edge list — oracle/ref_path.py).  Block level: the reference's
    model classes cannot use this extractor (forward_encode indexes its output with [0], nn/nets.py:134-135)."""
    import net_blocks as ref_blocks
    cfg = {'EConv_hidden': 32, 'EConv_feature': 24}
    torch.manual_seed(1100)
    net = ref_blocks.PointNetPlusPlus(16, dict(cfg))
    net.train()
    g = torch.Generator().manual_seed(1101)
    pos = torch.randn(2, 160, 3, generator=g) * 0.45
    wgt = torch.randn(2, 16, generator=g)
    state0 = copy.deepcopy(net.state_dict())
    torch.manual_seed(1102)                      # the forward draws the random fps start points
    out = net(pos)
    (out * wgt).sum().backward()
    conv = net.sa1_module.conv
    fx = {'config': cfg, 'out_size': 16, 'seed': 1100, 'fwd_seed': 1102, 'positions': pos, 'wgt': wgt, 'out': out.detach().clone(),
          'provenance': 'reference class on restated fps / radius / PointConv with PyG conventions (random fps start from '
                        'torch.rand, add_self_loops re-indexing)',
          'edge_index': conv.last_edge_index.clone(),
          'state_dict': state0, 'grads': {n: p.grad.clone() for n, p in net.named_parameters()},
          'state_keys': [(k, tuple(v.shape)) for k, v in state0.items()], 'torch': torch.__version__, 'threads': 1}
    path = os.path.join(REPO, 'tests', 'golden', 'pointnetpp_small.pt')
    torch.save(fx, path)
    print('%-28s |out| max %.4f  %.1f KB' % ('pointnetpp_small', out.abs().max().item(), os.path.getsize(path) / 1024))


def run_stitch_known_answer():
    """The only trained weights the reference ships (models/att/neural_tailor_stitch_model.pth): StitchOnEdge3DPairs =
    MLP([16, 200, 200, 200, 1]) with real BatchNorm running statistics.  Eval-mode outputs of the reference's own class
    on seeded pair features are a true known-answer test for the eval path of the dense-MLP kernels; a training-mode
    fwd/bwd (batch statistics) from the same weights is stored as well."""
    with open(os.path.join(REF, 'models/att/stitch_model.yaml')) as f:
        cfg = yaml.safe_load(f)
    data_config = dict(cfg['dataset'])
    ckpt = torch.load(os.path.join(REF, 'models/att/neural_tailor_stitch_model.pth'), map_location='cpu',
                      weights_only=False)
    sd = {k[len('module.'):]: v.clone() for k, v in ckpt['model_state_dict'].items()}
    nn_cfg = copy.deepcopy(cfg['NN'])
    model = ref_nets.StitchOnEdge3DPairs(data_config, copy.deepcopy(nn_cfg), copy.deepcopy(nn_cfg.get('loss', {})))
    model.load_state_dict(sd)
    g = torch.Generator().manual_seed(4242)
    pairs = torch.randn(6, 50, data_config['element_size'], generator=g)
    labels = (torch.rand(6, 50, generator=g) < 0.2)
    model.eval()
    with torch.no_grad():
        out_eval = model(pairs).clone()
    model.train()
    model.loss.with_quality_eval = False
    out_train = model(pairs)
    loss, loss_dict, _ = model.loss(out_train, labels)
    loss.backward()
    fx = {'model': 'StitchOnEdge3DPairs', 'data_config': {'element_size': data_config['element_size']},
          'nn_config': {k: nn_cfg[k] for k in ('stitch_hidden_size', 'stitch_mlp_n_layers') if k in nn_cfg},
          'state_dict': sd, 'pairs': pairs, 'labels': labels, 'out_eval': out_eval,
          'out_train': out_train.detach().clone(), 'loss': loss.detach().clone(),
          'grads': {n: p.grad.clone() for n, p in model.named_parameters()},
          'state_after_train': {k: v.clone() for k, v in model.state_dict().items()},
          'torch': torch.__version__, 'threads': 1}
    out = os.path.join(REPO, 'tests', 'golden', 'stitch_pairs_known_answer.pt')
    torch.save(fx, out)
    print('%-28s eval |out| max %.4f  loss=%.6f  %.1f KB' % ('stitch_pairs_known_answer', out_eval.abs().max().item(),
                                                         loss.item(), os.path.getsize(out) / 1024))


if __name__ == '__main__':
    os.makedirs(os.path.join(REPO, 'tests', 'golden'), exist_ok=True)
    lstm_yaml, att_yaml = 'models/baseline/lstm_stitch_tags.yaml', 'models/att/att.yaml'
    run_case('GarmentFullPattern3D', lstm_yaml, SMALL_NN, 2, 64, 100, 'full3d_small', True)
    run_case('GarmentSegmentPattern3D', att_yaml, SMALL_NN, 2, 64, 200, 'segment3d_small', True)
    # shipped hyper-parameters; weights are re-derivable from the seed, so only results are stored
    run_case('GarmentFullPattern3D', lstm_yaml, {}, 2, 128, 300, 'full3d_shipped', False)
    run_case('GarmentSegmentPattern3D', att_yaml, {}, 2, 128, 400, 'segment3d_shipped', False)
    run_case('GarmentFullPattern3D', lstm_yaml, {'k_neighbors': 16}, 2, 256, 500, 'full3d_k16', False)
    # ---- round 2 ------------------------------------------------------------------------------------------------
    # cfg 4's neighbourhood size (k = 20 > 16: the paired fallback kernel) at model level, shipped attention YAML
    run_case('GarmentSegmentPattern3D', att_yaml, {'k_neighbors': 20}, 2, 256, 600, 'segment3d_k20', False)
    # global attention (local_attention False -> 403-wide point MLP; old checkpoints rely on it, nn/nets.py:210-216)
    run_case('GarmentSegmentPattern3D', att_yaml, {'local_attention': False}, 2, 128, 700, 'segment3d_globalatt', False)
    run_case('GarmentSegmentPattern3D', att_yaml, dict(SMALL_NN, local_attention=False), 2, 64, 710,
             'segment3d_globalatt_small', True)
    # the model's OWN default loss config: panel-origin matching on (nn/nets.py:83-97), and order matching on top
    run_case('GarmentFullPattern3D', lstm_yaml, SMALL_NN, 2, 64, 800, 'full3d_originmatch', True,
             loss_override={'panel_origin_invariant_loss': True})
    run_case('GarmentFullPattern3D', lstm_yaml, SMALL_NN, 2, 64, 810, 'full3d_ordermatch', True,
             loss_override={'panel_origin_invariant_loss': True, 'panel_order_inariant_loss': True,
                            'order_by': 'shape_translation'}, gt_extra=True)
    run_case('GarmentFullPattern3D', lstm_yaml, SMALL_NN, 2, 64, 820, 'full3d_ordermatch_placement', False,
             loss_override={'panel_origin_invariant_loss': False, 'panel_order_inariant_loss': True,
                            'order_by': 'placement'}, gt_extra=True)
    # alternative blocks behind the same YAML keys (nn/net_blocks.py:121-152,273-298,405-497)
    run_case('GarmentFullPattern3D', lstm_yaml, dict(SMALL_NN, global_pool='max'), 2, 64, 900, 'full3d_poolmax', True)
    run_case('GarmentFullPattern3D', lstm_yaml, dict(SMALL_NN, global_pool='add'), 2, 64, 910, 'full3d_pooladd', True)
    run_case('GarmentFullPattern3D', lstm_yaml, dict(SMALL_NN, EConv_aggr='mean'), 2, 64, 920, 'full3d_aggrmean', True)
    run_case('GarmentFullPattern3D', lstm_yaml, dict(SMALL_NN, EConv_aggr='add'), 2, 64, 930, 'full3d_aggradd', True)
    run_case('GarmentFullPattern3D', lstm_yaml, dict(SMALL_NN, EConv_hidden_depth=1), 2, 64, 940, 'full3d_depth1', True)
    run_case('GarmentFullPattern3D', lstm_yaml, dict(SMALL_NN, EConv_hidden_depth=3), 2, 64, 950, 'full3d_depth3', True)
    run_case('GarmentFullPattern3D', lstm_yaml, dict(SMALL_NN, panel_decoder='MLPDecoder', panel_n_layers=2,
                                                     pattern_decoder='MLPDecoder', pattern_n_layers=1),
             2, 64, 960, 'full3d_mlpdec', False)
    run_case('GarmentFullPattern3D', lstm_yaml, dict(SMALL_NN, panel_decoder='GRUDecoderModule',
                                                     pattern_decoder='GRUDecoderModule'), 2, 64, 970, 'full3d_gru', True)
    run_case('GarmentFullPattern3D', lstm_yaml, dict(SMALL_NN, panel_decoder='LSTMDoubleReverseDecoderModule',
                                                     pattern_decoder='LSTMDoubleReverseDecoderModule'),
             2, 64, 980, 'full3d_lstm2rev', True)
    run_case('GarmentSegmentPattern3D', att_yaml, dict(SMALL_NN, global_pool='max'), 2, 64, 990, 'segment3d_poolmax', True)
    run_case('GarmentSegmentPattern3D', att_yaml, dict(SMALL_NN, global_pool='add'), 2, 64, 995, 'segment3d_pooladd', True)
    # ---- round 3: the stitch terms of the benchmarked model's own YAML (active from epoch_with_stitches = 40 on:
    # nn/metrics/composed_loss.py:259-266,336-362; nn/metrics/losses.py:54-180) and the re-numbering of the stitched edges by
    # the order / origin matching (:505-517,604-617,705-755)
    run_case('GarmentFullPattern3D', lstm_yaml, SMALL_NN, 3, 64, 1200, 'full3d_stitch', True, epoch=40, stitch_gt=True)
    run_case('GarmentFullPattern3D', lstm_yaml, SMALL_NN, 3, 64, 1210, 'full3d_stitch_match', True, epoch=40, stitch_gt=True,
             loss_override={'panel_origin_invariant_loss': True, 'panel_order_inariant_loss': True,
                            'order_by': 'shape_translation'}, gt_extra=True)
    run_case('GarmentFullPattern3D', lstm_yaml, SMALL_NN, 3, 64, 1220, 'full3d_stitch_hardnet', False, epoch=41,
             stitch_gt=True, gt_extra=True,
             loss_override={'stitch_hardnet_version': True, 'panel_order_inariant_loss': True, 'order_by': 'stitches',
                            'stitch_supervised_weight': 0.1,
                            'loss_components': ['shape', 'loop', 'rotation', 'translation', 'stitch', 'stitch_supervised',
                                                'free_class']})
    # recurrent dropout (nn/nets.py:67,113,123 -> nn.LSTM / nn.GRU(dropout=p): a Bernoulli mask between the layers, drawn on
    # the CPU generator after the random start states)
    run_case('GarmentFullPattern3D', lstm_yaml, dict(SMALL_NN, dropout=0.25), 2, 64, 1300, 'full3d_dropout', True)
    run_case('GarmentFullPattern3D', lstm_yaml, dict(SMALL_NN, dropout=0.4, panel_decoder='GRUDecoderModule',
                                                     pattern_decoder='LSTMDoubleReverseDecoderModule'),
             2, 64, 1310, 'full3d_dropout_gru_2rev', True)
    # EdgeConv widths outside the fused kernels' menu (EConv_hidden not a multiple of 4 / wider than 256)
    run_case('GarmentFullPattern3D', lstm_yaml, dict(SMALL_NN, EConv_hidden=30, EConv_feature=22), 2, 64, 1400,
             'full3d_hidden30', True)
    run_case('GarmentSegmentPattern3D', att_yaml, dict(SMALL_NN, EConv_hidden=260), 2, 64, 1410, 'segment3d_hidden260', False)
    # the segmentation term on the attention weights (composed_loss.py:323-332; entmax stub -> PARITY UNPINNED arithmetic)
    run_case('GarmentSegmentPattern3D', att_yaml, SMALL_NN, 2, 64, 1500, 'segment3d_segmloss', True,
             loss_override={'loss_components': ['shape', 'loop', 'rotation', 'translation', 'segmentation']})
    run_case('GarmentSegmentPattern3D', att_yaml, {}, 2, 128, 1510, 'segment3d_segmloss_shipped', False,
             loss_override={'loss_components': ['shape', 'loop', 'rotation', 'translation', 'segmentation'],
                            'panel_origin_invariant_loss': True})
    if not ONLY or 'stitch_pairs_known_answer' in ONLY:
        run_stitch_known_answer()
    if not ONLY or 'pointnetpp_small' in ONLY:
        run_pointnet_case()
