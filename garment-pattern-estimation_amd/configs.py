"""Hyper-parameters of the two shipped reference models, as data (values from
/root/reference/models/baseline/lstm_stitch_tags.yaml:40-139 and models/att/att.yaml:40-139; the reference tree does
not travel to the GPU box).  `data_config` is the YAML's `dataset` section with max_pattern_len forced to 23, which is
what the dataset does when `panel_classification` is set (nn/data/datasets.py:377-379)."""
import copy

_STANDARDIZE = {
    'f_scale': [16.351303100585938, 30.945703506469727, 9.60141944885254],
    'f_shift': [0.037076108157634735, -28.06070327758789, 1.0775548219680786],
    'gt_scale': {
        'outlines': [25.267892837524418, 31.298505783081055, 0.2677369713783264, 0.2352069765329361],
        'rotations': [1.7071068286895752, 1.9238795042037964, 1.7071068286895752, 1],
        'stitch_tags': [119.98278045654295, 156.0384521484375, 105.92605590820312],
        'translations': [109.58930206298828, 98.27909088134766, 37.84679412841797]},
    'gt_shift': {
        'outlines': [0, 0, 0.14890235662460327, 0.05642016604542732],
        'rotations': [-0.7071067690849304, -0.9238795042037964, -1, 0],
        'stitch_tags': [-59.99139022827149, -78.12358856201172, -52.95616912841797],
        'translations': [-55.255470275878906, -20.001333236694336, -17.086795806884766]}}

_DATA = {
    'mesh_samples': 2000, 'max_pattern_len': 23, 'max_panel_len': 14, 'max_num_stitches': 24,
    'element_size': 4, 'rotation_size': 4, 'translation_size': 3, 'explicit_stitch_tags': False,
    'stitch_tag_size': 3, 'point_noise_w': 0, 'standardize': _STANDARDIZE}

_NN_COMMON = {
    'feature_extractor': 'EdgeConvFeatures', 'conv_depth': 2, 'k_neighbors': 5, 'EConv_hidden': 200,
    'EConv_hidden_depth': 2, 'EConv_feature': 150, 'EConv_aggr': 'max', 'global_pool': 'mean',
    'graph_pooling': False, 'pool_ratio': 0.1, 'local_attention': True,
    'panel_decoder': 'LSTMDecoderModule', 'panel_encoding_size': 250, 'panel_hidden_size': 250,
    'panel_n_layers': 3, 'lstm_init': 'kaiming_normal_',
    'pattern_decoder': 'LSTMDecoderModule', 'pattern_encoding_size': 250, 'pattern_hidden_size': 250,
    'pattern_n_layers': 2, 'stitch_tag_dim': 3}

_LOSS_COMMON = {
    'stitch_tags_margin': 0.3, 'stitch_hardnet_version': False, 'loop_loss_weight': 1.,
    'segm_loss_weight': 0.05, 'epoch_with_stitches': 40, 'panel_origin_invariant_loss': False,
    'panel_order_inariant_loss': False, 'epoch_with_order_matching': 0, 'order_by': 'shape_translation'}


def data_config():
    return copy.deepcopy(_DATA)


def lstm_model_config(**override):
    """models/baseline/lstm_stitch_tags.yaml  (GarmentFullPattern3D)."""
    nn_cfg = dict(_NN_COMMON, model='GarmentFullPattern3D', skip_connections=False)
    nn_cfg['loss'] = dict(_LOSS_COMMON,
                          loss_components=['shape', 'loop', 'rotation', 'translation', 'stitch', 'free_class'],
                          quality_components=['shape', 'discrete', 'rotation', 'translation', 'stitch',
                                              'free_class'])
    nn_cfg.update(override)
    return copy.deepcopy(nn_cfg)


def att_model_config(**override):
    """models/att/att.yaml  (GarmentSegmentPattern3D)."""
    nn_cfg = dict(_NN_COMMON, model='GarmentSegmentPattern3D', skip_connections=True)
    nn_cfg['loss'] = dict(_LOSS_COMMON, loss_components=['shape', 'loop', 'rotation', 'translation'],
                          quality_components=['shape', 'discrete', 'rotation', 'translation'])
    nn_cfg.update(override)
    return copy.deepcopy(nn_cfg)
