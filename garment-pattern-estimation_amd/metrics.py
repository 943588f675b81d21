"""Loss terms that seed the backward pass — the reference's nn/metrics/composed_loss.py:129-334,428-703 and
nn/metrics/losses.py:8-51 behind the same classes, config keys, call signature and loss-dict keys.

On device tensors (the only thing a model of this package can produce) everything runs in libgpe_hip.so:
  * shape / loop / rotation / translation terms, value and gradient: ONE forward and ONE backward launch
    (ops.PatternLossFn) — the reference's PanelLoopLoss walks B*23 panels in a Python loop with a tensor-valued `if` per
    panel, i.e. one host sync per panel on a GPU;
  * `panel_origin_invariant_loss` (the default of GarmentFullPattern3D's own loss config): ops.origin_match, one wave per
    panel instead of a Python loop over panels x edge shifts;
  * `panel_order_inariant_loss`: ops.order_match, the greedy assignment of composed_loss.py:530-570 as one workgroup per
    pattern; the random pre-matching permutation (epoch < epoch_with_order_matching) and the gathers stay torch calls.
  * from `epoch_with_stitches` on, the stitch terms of the shipped YAMLs (composed_loss.py:336-362): PatternStitchLoss
    (losses.py:54-180, both negative-term variants), supervised stitch tags, free-edge classification — ops.StitchLossFn,
    one forward and one backward launch — and the re-numbering of the stitched edges / shift of the free-edge mask that the
    two matchings imply (ops.stitch_renumber, ops.panel_shift).
There is no CPU or torch-math path in `ComposedPatternLoss`: predictions must be fp32 device tensors (what the models of
this package produce); anything else raises, like the model path itself.

The segmentation term (entmax.SparsemaxLoss on the attention weights, composed_loss.py:323-332) runs through
ops.SparsemaxLossFn (entmax is third-party and un-vendored: its published loss restated).  Quality components are
evaluation-side bookkeeping: `with_quality_eval` is accepted and ignored, no quality keys are added to the loss dict."""
import torch
import torch.nn as nn


def eval_pad_vector(data_stats={}):
    if data_stats:
        shift = torch.Tensor(data_stats['shift'])
        scale = torch.Tensor(data_stats['scale'])
        return -shift / scale
    return None


class PanelLoopLoss:
    """nn/metrics/losses.py:8-51 (panels with < 3 edges contribute zero: the reference `continue`s) through the HIP loss
    kernel (ops.PatternLossFn with only the loop term switched on).  ComposedPatternLoss evaluates the loop term inside its one
    fused launch; this class exists for callers that use the reference's class on its own.  Device tensors only."""

    def __init__(self, max_edges_in_panel, data_stats={}):
        self.data_stats = data_stats
        self.pad_vector = eval_pad_vector(data_stats)

    def __call__(self, predicted_panels, gt_panel_num_edges=None):
        from . import ops
        if self.pad_vector is None:
            raise ValueError('PanelLoopLoss needs data_stats (the reference would fail on a None pad vector too)')
        if not predicted_panels.is_cuda or predicted_panels.dtype != torch.float32:
            raise RuntimeError('PanelLoopLoss runs on fp32 device tensors only (got %s / %s); there is no CPU path'
                               % (predicted_panels.device, predicted_panels.dtype))
        x = predicted_panels if predicted_panels.dim() > 3 else predicted_panels.unsqueeze(1)      # [B, P, L, D]
        if x.dim() != 4 or x.shape[-1] < 4:
            raise ValueError('PanelLoopLoss expects panels of [.., L, >= 4] edge features (got %s)' % (tuple(predicted_panels.shape),))
        x = x[..., :4]
        B, P, L = x.shape[:3]
        if gt_panel_num_edges is None:
            n = torch.full((B * P,), L, device=x.device, dtype=torch.int32)
        else:
            n = gt_panel_num_edges.to(x.device).int().reshape(-1).contiguous()
        out = ops.PatternLossFn.apply(x, None, None, x.detach(), None, None, n, ops.LOSS_LOOP,
                                      float(self.pad_vector[0]), float(self.pad_vector[1]), 1.0)
        return out[0]                      # loop weight 1: the total IS the loop term (element 0 carries the gradient)


class ComposedLoss:
    """nn/metrics/composed_loss.py:11-128 (the base used by StitchOnEdge3DPairs): BCE-with-logits on edge pairs.
    Caller-side torch ops (§8f rank 4 row; the model's MLP is the kernel path)."""

    def __init__(self, data_config, in_config={}):
        self.config = {'loss_components': [], 'quality_components': []}
        self.config.update(in_config)
        self.with_quality_eval = True
        self.training = False
        self.l_components = self.config['loss_components']
        self.q_components = self.config['quality_components']
        if 'edge_pair_class' in self.l_components:
            self.bce_logits_loss = nn.BCEWithLogitsLoss()

    def __call__(self, preds, ground_truth, names=None, epoch=1000):
        self.device = preds.device
        ground_truth = ground_truth.to(self.device)
        full_loss, loss_dict = 0., {}
        if 'edge_pair_class' in self.l_components:
            pair_loss = self.bce_logits_loss(preds.view(-1), ground_truth.view(-1).float())
            loss_dict.update(edge_pair_class_loss=pair_loss)
            full_loss = full_loss + pair_loss
        if self.with_quality_eval:
            with torch.no_grad():
                cls = torch.round(torch.sigmoid(preds))
                if 'edge_pair_class' in self.q_components:
                    loss_dict.update(edge_pair_class_acc=(cls == ground_truth).sum().float() / ground_truth.numel())
                if 'edge_pair_stitch_recall' in self.q_components:
                    hit = ((cls == 1) & (ground_truth == 1)).sum().float()
                    n_pred, n_gt = (cls == 1).sum().float(), (ground_truth == 1).sum().float()
                    loss_dict.update(stitch_precision=hit / n_pred if n_pred else 0,
                                     stitch_recall=hit / n_gt if n_gt else 0)
        return full_loss, loss_dict, False

    def eval(self):
        self.training = False

    def train(self, mode=True):
        self.training = mode


class ComposedPatternLoss:
    """Same constructor / call signature / config keys / loss-dict keys as the reference class."""

    def __init__(self, data_config, in_config={}):
        self.config = {
            'loss_components': ['shape'], 'quality_components': [], 'loop_loss_weight': 1.,
            'segm_loss_weight': 0.05, 'stitch_tags_margin': 0.3, 'epoch_with_stitches': 40,
            'stitch_supervised_weight': 0.1, 'stitch_hardnet_version': False,
            'panel_origin_invariant_loss': True, 'panel_order_inariant_loss': True,
            'order_by': 'placement', 'epoch_with_order_matching': 0}
        self.config.update(in_config)
        self.with_quality_eval = True
        self.training = False
        self.debug_prints = False
        self.l_components = self.config['loss_components']
        self.q_components = self.config['quality_components']
        self.max_panel_len = data_config['max_panel_len']
        self.max_pattern_size = data_config['max_pattern_len']
        data_stats = data_config['standardize']
        self.gt_outline_stats = {'shift': data_stats['gt_shift']['outlines'],
                                 'scale': data_stats['gt_scale']['outlines']}
        self.cluster_resolution_mapping = {}
        if any(c in self.l_components for c in ('shape', 'rotation', 'translation')):
            self.regression_loss = nn.MSELoss()
        if 'loop' in self.l_components:
            self.loop_loss = PanelLoopLoss(self.max_panel_len, data_stats=self.gt_outline_stats)
        self.last_permutation = None          # [B, P] int64 of the last order matching (diagnostics / tests)
        self.last_leading_edges = None        # [B*P] int32 of the last origin matching

    # ---- helpers ---------------------------------------------------------------------------------------------
    def _stitch_terms_active(self, epoch):
        return epoch >= self.config['epoch_with_stitches'] and any(
            c in self.l_components for c in ('stitch', 'stitch_supervised', 'free_class'))

    @staticmethod
    def _feature_permute(feat, perm):
        """composed_loss.py:572-590: gather along the panel dimension."""
        ext = perm
        while ext.dim() < feat.dim():
            ext = ext.unsqueeze(-1)
        return torch.gather(feat, 1, ext.expand(feat.shape))

    def _order_features(self, preds, gt):
        """composed_loss.py:437-488: which features drive the panel-order matching."""
        by = self.config['order_by']
        if by == 'placement':
            if 'translations' not in preds or 'rotations' not in preds:
                raise ValueError('ComposedPatternLoss::Error::Ordering by placement requested but placement is not predicted')
            return (torch.cat([preds['translations'], preds['rotations']], dim=-1),
                    torch.cat([gt['translations'], gt['rotations']], dim=-1))
        if by == 'translation':
            if 'translations' not in preds:
                raise ValueError('ComposedPatternLoss::Error::Ordering by translation requested but translation is not predicted')
            return preds['translations'], gt['translations']
        if by == 'shape_translation':
            if 'translations' not in preds:
                raise ValueError('ComposedPatternLoss::Error::Ordering by translation requested but translation is not predicted')
            B, P = preds['outlines'].shape[:2]
            return (torch.cat([preds['translations'], preds['outlines'].contiguous().view(B, P, -1)], dim=-1),
                    torch.cat([gt['translations'], gt['outlines'].contiguous().view(B, P, -1)], dim=-1))
        if by == 'stitches':
            if 'free_edges_mask' not in preds or 'translations' not in preds or 'rotations' not in preds:
                raise ValueError('ComposedPatternLoss::Error::Ordering by stitches requested but free edges mask or placement are not predicted')
            pf = torch.cat([preds['translations'], preds['rotations']], dim=-1)
            gf = torch.cat([gt['translations'], gt['rotations']], dim=-1)
            if self.epoch >= self.config['epoch_with_stitches']:          # composed_loss.py:464-477
                B, P = preds['free_edges_mask'].shape[:2]
                pf = torch.cat([pf, torch.round(torch.sigmoid(preds['free_edges_mask'])).reshape(B, P, -1)], dim=-1)
                gf = torch.cat([gf, gt['free_edges_mask'].reshape(B, P, -1).to(gf.dtype)], dim=-1)
            else:
                print('ComposedPatternLoss::Warning::skipped order match by stitch tags as stitch loss is not enabled')
            return pf, gf
        raise NotImplementedError(
            'ComposedPatternLoss::Error::Ordering by requested feature <{}> is not implemented'.format(by))

    def _panel_order_match(self, pred_feat, gt_feat):
        """composed_loss.py:530-570."""
        B, P = pred_feat.shape[0], gt_feat.shape[1]
        if self.epoch < self.config['epoch_with_order_matching']:
            return torch.stack([torch.randperm(P, dtype=torch.long, device=pred_feat.device) for _ in range(B)])
        pf = pred_feat.detach().reshape(B, P, -1).float()
        gf = gt_feat.reshape(B, P, -1).float()
        if pf.is_cuda:
            from . import ops
            perm, fail = ops.order_match(pf, gf)
            self._order_fail = fail              # device flag; checked lazily (no host sync on the hot path)
            return perm
        raise RuntimeError('ComposedPatternLoss runs on the MI355X only (got %s predictions); there is no CPU path' % pf.device)

    def check_order_match(self):
        """Host-side check of the last device order matching (the reference raises inside the loss; here the flag is read
        only when asked, so the training step stays free of host syncs)."""
        fail = getattr(self, '_order_fail', None)
        if fail is not None and int(fail.item()):
            raise ValueError('ComposedPatternLoss::Error::Failed to match panel order')

    def _gt_order_match(self, preds, gt):
        """composed_loss.py:428-528."""
        with torch.no_grad():
            pf, gf = self._order_features(preds, gt)
            perm = self._panel_order_match(pf, gf)
            self.last_permutation = perm
            out = dict(gt)
            out['outlines'] = self._feature_permute(gt['outlines'], perm)
            out['num_edges'] = self._feature_permute(gt['num_edges'], perm)
            if 'empty_panels_mask' in gt:
                out['empty_panels_mask'] = self._feature_permute(gt['empty_panels_mask'], perm)
            if 'rotation' in self.l_components:
                out['rotations'] = self._feature_permute(gt['rotations'], perm)
            if 'translation' in self.l_components:
                out['translations'] = self._feature_permute(gt['translations'], perm)
            if self._stitch_terms_active(self.epoch):                      # composed_loss.py:505-517
                from . import ops
                self._need_device(gt['stitches'])
                out['stitches'] = ops.stitch_renumber(gt['stitches'].long().contiguous(),
                                                      gt['num_stitches'].long().contiguous(), self.max_pattern_size,
                                                      self.max_panel_len, perm=perm.contiguous())
                out['free_edges_mask'] = self._feature_permute(gt['free_edges_mask'], perm)
                if 'stitch_supervised' in self.l_components:
                    out['stitch_tags'] = self._feature_permute(gt['stitch_tags'], perm)
        return out

    @staticmethod
    def _need_device(t):
        if not t.is_cuda:
            raise RuntimeError('ComposedPatternLoss runs on the MI355X only (got a %s tensor); there is no CPU path' % t.device)

    def _rotate_gt(self, preds, gt, gt_num_edges):
        """composed_loss.py:593-623,656-755."""
        with torch.no_grad():
            out = dict(gt)
            ol, gto = preds['outlines'], gt['outlines']
            if ol.is_cuda:
                from . import ops
                ne = gt_num_edges.contiguous()
                out['outlines'], lead = ops.origin_match(ol, gto.float().contiguous(), ne)
                self.last_leading_edges = lead
                if self._stitch_terms_active(self.epoch):                  # composed_loss.py:604-617
                    out['stitches'] = ops.stitch_renumber(gt['stitches'].long().contiguous(),
                                                          gt['num_stitches'].long().contiguous(), self.max_pattern_size,
                                                          self.max_panel_len, lead=lead, num_edges=ne)
                    out['free_edges_mask'] = ops.panel_shift(gt['free_edges_mask'].float(), lead, ne)
                    if 'stitch_supervised' in self.l_components:
                        out['stitch_tags'] = ops.panel_shift(gt['stitch_tags'].float(), lead, ne)
            else:
                self._need_device(ol)
        return out

    # ---- main entry ------------------------------------------------------------------------------------------
    def __call__(self, preds, ground_truth, names=None, epoch=1000):
        self.device = preds['outlines'].device
        self.epoch = epoch
        self._need_device(preds['outlines'])
        for key in ground_truth:
            ground_truth[key] = ground_truth[key].to(self.device)
        gt = ground_truth
        if self.config['panel_order_inariant_loss']:
            if 'segmentation' in self.l_components:                            # composed_loss.py:242-243
                raise NotImplementedError('Order matching not supported for training with segmentation losses')
            gt = self._gt_order_match(preds, gt)
        gt_num_edges = gt['num_edges'].int().view(-1)
        if self.config['panel_origin_invariant_loss']:
            gt = self._rotate_gt(preds, gt, gt_num_edges)
        full_loss, loss_dict = self._main_losses(preds, gt, gt_num_edges)
        if 'segmentation' in self.l_components:                                # composed_loss.py:323-332
            from . import ops
            att = preds['att_weights']
            segm = ops.SparsemaxLossFn.apply(att.reshape(-1, att.shape[-1]), gt['segmentation'].reshape(-1))
            full_loss = full_loss + self.config['segm_loss_weight'] * segm
            loss_dict.update(segm_loss=segm)
        if self._stitch_terms_active(epoch):
            self.last_matched_stitch_gt = {k: gt[k] for k in ('stitches', 'free_edges_mask') if k in gt}   # diagnostics / tests
            extra, extra_dict = self._stitch_losses(preds, gt)
            full_loss = full_loss + extra
            loss_dict.update(extra_dict)
        loss_update_ind = (
            epoch == self.config['epoch_with_stitches'] and any(
                el in self.l_components for el in ['stitch', 'stitch_supervised', 'free_class'])
            or epoch == self.config['epoch_with_order_matching'] and self.config['panel_order_inariant_loss'])
        return full_loss, loss_dict, loss_update_ind

    def _stitch_losses(self, preds, gt):
        """composed_loss.py:336-362 through ops.StitchLossFn."""
        from . import ops
        comps = self.l_components
        flags = (ops.STITCH_MAIN if 'stitch' in comps else 0) | (ops.STITCH_FREE if 'free_class' in comps else 0) | \
                (ops.STITCH_SUP if 'stitch_supervised' in comps else 0)
        if 'stitch' in comps and self.config['stitch_hardnet_version']:
            flags |= ops.STITCH_HARDNET
        tags = preds['stitch_tags'] if flags & (ops.STITCH_MAIN | ops.STITCH_SUP) else None
        logits = preds['free_edges_mask'] if flags & ops.STITCH_FREE else None
        out = ops.StitchLossFn.apply(
            tags, logits,
            gt['stitches'].long().contiguous() if flags & ops.STITCH_MAIN else None,
            gt['num_stitches'].long().contiguous() if flags & ops.STITCH_MAIN else None,
            gt['free_edges_mask'].float().contiguous() if flags & ops.STITCH_FREE else None,
            gt['stitch_tags'].float().contiguous() if flags & ops.STITCH_SUP else None,
            flags, float(self.config['stitch_tags_margin']), float(self.config['stitch_supervised_weight']))
        loss_dict = {}
        if 'stitch' in comps:
            loss_dict.update(stitch_similarity_loss=out[1], stitch_neg_loss=out[2])
        if 'stitch_supervised' in comps:
            loss_dict.update(stitch_supervised_loss=out[3])
        if 'free_class' in comps:
            loss_dict.update(free_edges_loss=out[4])
        return out[0], loss_dict

    def _main_losses(self, preds, gt, gt_num_edges):
        """composed_loss.py:294-321."""
        comps = self.l_components
        ol = preds['outlines']
        if ol.is_cuda and ol.dtype == torch.float32:
            from . import ops
            flags = (ops.LOSS_SHAPE if 'shape' in comps else 0) | (ops.LOSS_LOOP if 'loop' in comps else 0) | \
                    (ops.LOSS_ROT if 'rotation' in comps else 0) | (ops.LOSS_TR if 'translation' in comps else 0)
            pad0 = pad1 = 0.0
            if 'loop' in comps:
                if self.loop_loss.pad_vector is None:
                    raise ValueError('PanelLoopLoss needs data_stats')
                pv = self.loop_loss.pad_vector
                if getattr(self, '_pad_cache', (None,))[0] is not pv:          # (two tensor -> float reads per step otherwise)
                    self._pad_cache = (pv, float(pv[0]), float(pv[1]))
                pad0, pad1 = self._pad_cache[1], self._pad_cache[2]
            rot = preds['rotations'] if 'rotation' in comps else None
            tr = preds['translations'] if 'translation' in comps else None
            out = ops.PatternLossFn.apply(
                ol, rot, tr, gt['outlines'].float().contiguous(),
                gt['rotations'].float().contiguous() if rot is not None else None,
                gt['translations'].float().contiguous() if tr is not None else None,
                gt_num_edges.contiguous(), flags, pad0, pad1, float(self.config['loop_loss_weight']))
            loss_dict = {}
            if 'shape' in comps:
                loss_dict.update(pattern_loss=out[1])
            if 'loop' in comps:
                loss_dict.update(loop_loss=out[2])
            if 'rotation' in comps:
                loss_dict.update(rotation_loss=out[3])
            if 'translation' in comps:
                loss_dict.update(translation_loss=out[4])
            return out[0], loss_dict
        raise RuntimeError('ComposedPatternLoss runs on fp32 device tensors only (got %s / %s); there is no CPU path'
                           % (ol.device, ol.dtype))

    def eval(self):
        self.training = False

    def train(self, mode=True):
        self.training = mode
