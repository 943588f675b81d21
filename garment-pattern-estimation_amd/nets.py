"""Drop-in for the reference's nn/nets.py: same class names (resolved with getattr from the YAML, nn/train.py:120,
nn/experiment.py:232), constructor signature `(data_config, config={}, in_loss_config={})`, merged `.config`,
`forward(positions_batch, **kwargs) -> dict`, `.loss(preds, gt, epoch=)`, parameter/buffer names and shapes."""
import torch
import torch.nn as nn

from . import net_blocks as blocks
from . import ops
from .metrics import ComposedLoss, ComposedPatternLoss


class BaseModule(nn.Module):
    """nn/nets.py:11-37."""

    def __init__(self):
        super().__init__()
        self.config = {'loss': 'MSELoss', 'model': self.__class__.__name__}
        self.regression_loss = nn.MSELoss()

    def loss(self, preds, ground_truth, **kwargs):
        ground_truth = ground_truth.to(preds.device)
        loss = self.regression_loss(preds, ground_truth)
        return loss, {'regression loss': loss}, False

    def train(self, mode=True):
        super().train(mode)
        if isinstance(self.loss, object):
            self.loss.train(mode)
        return self

    # ---- weight-derived kernel operands (ops.PackPlan): one refresh launch per optimizer step ----------------
    def _pack_modules(self):
        return [m for m in self.children() if hasattr(m, 'register_packs')]

    def _register_own_packs(self, plan):
        pass

    def refresh_packs(self):
        plan = self.__dict__.get('_pack_plan')
        if plan is None:
            plan = ops.PackPlan()
            for m in self._pack_modules():
                m.register_packs(plan)
            self._register_own_packs(plan)
            self.__dict__['_pack_plan'] = plan
        plan.refresh()

    def eval(self):
        super().eval()
        if isinstance(self.loss, object):
            self.loss.eval()
        return self


# ---- GarmentFullPattern3D: what the reference's constructor fixes (nn/nets.py:49-130), as tables ------------------------------------
# attribute <- data_config key (nn/nets.py:53-57)
_DATA_FIELDS = (('panel_elem_len', 'element_size'), ('max_panel_len', 'max_panel_len'), ('max_pattern_size', 'max_pattern_len'),
                ('rotation_size', 'rotation_size'), ('translation_size', 'translation_size'))
# NN defaults (nn/nets.py:60-73); a *_hidden_size the caller left out falls back to the matching *_encoding_size IN THE CALLER'S DICT
# (nn/nets.py:75-78: configs saved before the hidden sizes existed)
_NN_DEFAULTS = {'panel_encoding_size': 250, 'panel_hidden_size': 250, 'panel_n_layers': 3,
                'pattern_encoding_size': 250, 'pattern_hidden_size': 250, 'pattern_n_layers': 2,
                'dropout': 0, 'lstm_init': 'kaiming_normal_', 'feature_extractor': 'EdgeConvFeatures',
                'panel_decoder': 'LSTMDecoderModule', 'pattern_decoder': 'LSTMDecoderModule', 'stitch_tag_dim': 3}
_HIDDEN_FALLBACK = (('panel_hidden_size', 'panel_encoding_size'), ('pattern_hidden_size', 'pattern_encoding_size'))
# loss defaults (nn/nets.py:83-92)
_LOSS_DEFAULTS = {'loss_components': ['shape', 'loop', 'rotation', 'translation'],
                  'quality_components': ['shape', 'discrete', 'rotation', 'translation'],
                  'loop_loss_weight': 1., 'stitch_tags_margin': 0.3, 'epoch_with_stitches': 40,
                  'stitch_supervised_weight': 0.1, 'stitch_hardnet_version': False, 'panel_origin_invariant_loss': True}


class GarmentFullPattern3D(BaseModule):
    """nn/nets.py:41-184: EdgeConv encoder -> pattern LSTM -> panel LSTM + placement Linear."""

    def __init__(self, data_config, config={}, in_loss_config={}):
        super().__init__()
        for attr, key in _DATA_FIELDS:
            setattr(self, attr, data_config[key])
        for hidden, enc in _HIDDEN_FALLBACK:                 # mutates the caller's dict (and raises KeyError when neither key is
            if hidden not in config:                         # given), like the reference
                config[hidden] = config[enc]
        self.config.update(_NN_DEFAULTS)
        self.config.update(config)
        self.loss = ComposedPatternLoss(data_config, {**{k: (list(v) if isinstance(v, list) else v) for k, v in _LOSS_DEFAULTS.items()},
                                                      **in_loss_config})
        self.config['loss'] = self.loss.config               # the loss object owns the merged dict from here on
        cfg = self.config
        self.feature_extractor = getattr(blocks, cfg['feature_extractor'])(cfg['pattern_encoding_size'], cfg)
        cfg.update(getattr(self.feature_extractor, 'config', {}))
        # one row per sequence decoder: name, class key, output width, sequence length (`out_len` is swallowed by **kwargs in the
        # recurrent decoders and used by MLPDecoder: nn/net_blocks.py:273-298,365)
        for name, out_width, out_len in (('panel', self.panel_elem_len + cfg['stitch_tag_dim'] + 1, self.max_panel_len),
                                         ('pattern', cfg['panel_encoding_size'], self.max_pattern_size)):
            setattr(self, name + '_decoder', getattr(blocks, cfg[name + '_decoder'])(
                encoding_size=cfg[name + '_encoding_size'], hidden_size=cfg[name + '_hidden_size'], out_elem_size=out_width,
                n_layers=cfg[name + '_n_layers'], out_len=out_len, dropout=cfg['dropout'], custom_init=cfg['lstm_init']))
        self.placement_decoder = nn.Linear(cfg['panel_encoding_size'], self.rotation_size + self.translation_size)

    def forward_encode(self, positions_batch):
        if isinstance(self.feature_extractor, blocks.EdgeConvFeatures):
            return self.feature_extractor(positions_batch, want_batch=False)[0]
        return self.feature_extractor(positions_batch)[0]

    def forward_pattern_decode(self, garment_encodings):
        panel_encodings = self.pattern_decoder(garment_encodings, self.max_pattern_size)
        return panel_encodings.contiguous().view(-1, panel_encodings.shape[-1])

    def _as_pattern(self, flat, last_dims):
        """[B * panels, ...] rows of a decoder -> [B, panels, *last_dims, -1] (views of ONE tensor per decoder, as in the reference)"""
        return flat.contiguous().view(-1, self.max_pattern_size, *last_dims, flat.shape[-1])

    def forward_panel_decode(self, flat_panel_encodings, batch_size):
        """nn/nets.py:143-169: per panel the edge sequence [edges][outline | stitch tag | free-edge logit] and the placement
        [rotation | translation]; the output dict slices those two tensors."""
        edges = self._as_pattern(self.panel_decoder(flat_panel_encodings, self.max_panel_len), (self.max_panel_len,))
        place = self._as_pattern(ops.linear(flat_panel_encodings, self.placement_decoder.weight, self.placement_decoder.bias), ())
        assert edges.shape[0] == batch_size
        e, r = self.panel_elem_len, self.rotation_size
        return {'outlines': edges[..., :e], 'rotations': place[..., :r], 'translations': place[..., r:],
                'stitch_tags': edges[..., e:-1], 'free_edges_mask': edges[..., -1]}

    def forward_decode(self, garment_encodings):
        flat_panel_encodings = self.forward_pattern_decode(garment_encodings)
        return self.forward_panel_decode(flat_panel_encodings, garment_encodings.size(0))

    def _register_own_packs(self, plan):
        plan.add_linear(self.placement_decoder.weight)

    def forward(self, positions_batch, **kwargs):
        self.refresh_packs()
        return self.forward_decode(self.forward_encode(positions_batch))


class GarmentSegmentPattern3D(GarmentFullPattern3D):
    """nn/nets.py:187-299: per-point sparsemax attention over the EdgeConv features -> per-panel pooled encodings ->
    panel LSTM.  The per-panel Python loop of the reference is one reduce-GEMM per cloud here."""

    def __init__(self, data_config, config={}, in_loss_config={}):
        if 'loss_components' not in in_loss_config:     # mutates the caller's dict like the reference (:194-199)
            in_loss_config.update(loss_components=['shape', 'loop', 'rotation', 'translation'],
                                  quality_components=['shape', 'discrete', 'rotation', 'translation'])
        super().__init__(data_config, config, in_loss_config)
        self.save_att_weights = 'segmentation' in self.loss.config['loss_components']
        if 'local_attention' not in self.config:
            self.config['local_attention'] = False
        attention_input_size = self.feature_extractor.config['EConv_feature']
        if not self.config['local_attention']:
            attention_input_size += self.config['pattern_encoding_size']
        if self.config['skip_connections']:
            attention_input_size += 3
        # parameter container with the reference's key layout: point_segment_mlp.0.{i}.{0,2}.* ; index 1 is the
        # parameter-free Sparsemax of the reference's Sequential
        self.point_segment_mlp = nn.Sequential(
            blocks.MLP([attention_input_size, attention_input_size, attention_input_size, self.max_pattern_size]),
            nn.Identity())
        panel_att_out_size = self.feature_extractor.config['EConv_feature']
        if self.config['skip_connections']:
            panel_att_out_size += 3
        self.panel_dec_lin = nn.Linear(panel_att_out_size, self.feature_extractor.config['panel_encoding_size'])
        del self.pattern_decoder

    def forward_panel_enc_from_3d(self, positions_batch):
        batch_size = positions_batch.shape[0]
        init_pattern_encodings, point_features_flat, batch = self.feature_extractor(
            positions_batch, not self.config['local_attention'])
        num_points = point_features_flat.shape[0] // batch_size
        point_features_flat = point_features_flat.contiguous()
        if self.config['local_attention']:
            att_in = point_features_flat
        else:
            glob = init_pattern_encodings.unsqueeze(1).repeat(1, num_points, 1).view(
                [-1, init_pattern_encodings.shape[-1]])
            att_in = torch.cat([glob, point_features_flat], dim=-1)
        logits = ops.dense_mlp(att_in, self.point_segment_mlp[0], self.training)
        points_weights = ops.SparsemaxFn.apply(logits)
        # "same pool as in initial extractor" (nn/nets.py:271-272): mean / max / add over the weighted features
        pooled = ops.AttentionPoolFn.apply(points_weights, point_features_flat, batch_size, num_points,
                                           self.feature_extractor.global_pool.pool_mode)
        panel_encodings = ops.linear(pooled, self.panel_dec_lin.weight, self.panel_dec_lin.bias)
        panel_encodings = panel_encodings.view(batch_size, -1, panel_encodings.shape[-1])
        points_weights = points_weights.view(batch_size, -1, points_weights.shape[-1]) \
            if self.save_att_weights else []
        return panel_encodings, points_weights

    def _register_own_packs(self, plan):
        plan.add_linear(self.placement_decoder.weight)
        plan.add_linear(self.panel_dec_lin.weight)
        blocks = [self.point_segment_mlp[0][i] for i in range(len(self.point_segment_mlp[0]))]
        plan.add_linear(blocks[0][0].weight)
        for b in blocks[1:]:
            plan.add_linear(b[0].weight, fwd=False, bwd=True)

    def forward(self, positions_batch, **kwargs):
        self.refresh_packs()
        batch_size = positions_batch.shape[0]
        panel_encodings, att_weights = self.forward_panel_enc_from_3d(positions_batch)
        panels = self.forward_panel_decode(panel_encodings.view(-1, panel_encodings.shape[-1]), batch_size)
        if len(att_weights) > 0:
            panels.update(att_weights=att_weights)
        return panels


class StitchOnEdge3DPairs(BaseModule):
    """nn/nets.py:303-353: binary classifier on pairs of 3D edges — MLP([element_size, 200 x 3, 1]) on every pair row.
    The reference ships trained weights for it (models/att/neural_tailor_stitch_model.pth)."""

    def __init__(self, data_config, config={}, in_loss_config={}):
        super().__init__()
        self.pair_feature_len = data_config['element_size']
        self.config.update({'stitch_hidden_size': 200, 'stitch_mlp_n_layers': 3})
        self.config.update(config)
        self.config['loss'] = {
            'loss_components': ['edge_pair_class'],
            'quality_components': ['edge_pair_class', 'edge_pair_stitch_recall'],
            'panel_origin_invariant_loss': False, 'panel_order_inariant_loss': False}
        self.config['loss'].update(in_loss_config)
        self.loss = ComposedLoss(data_config, self.config['loss'])
        self.config['loss'] = self.loss.config
        mid_layers = [self.config['stitch_hidden_size']] * self.config['stitch_mlp_n_layers']
        self.mlp = blocks.MLP([self.pair_feature_len] + mid_layers + [1])

    def _register_own_packs(self, plan):
        mlp_blocks = [self.mlp[i] for i in range(len(self.mlp))]
        plan.add_linear(mlp_blocks[0][0].weight)
        for b in mlp_blocks[1:]:
            plan.add_linear(b[0].weight, fwd=False, bwd=True)

    def forward(self, pairs_batch, **kwargs):
        self.refresh_packs()
        self.device = pairs_batch.device
        self.batch_size = pairs_batch.size(0)
        return_shape = list(pairs_batch.shape)
        return_shape.pop(-1)
        out = ops.dense_mlp(pairs_batch.contiguous().view(-1, pairs_batch.shape[-1]), self.mlp, self.training)
        return out.view(return_shape)
