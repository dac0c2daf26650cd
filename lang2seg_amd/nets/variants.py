"""The reference's ResNet network variants: one nets/network_*.py + nets/resnet_v1_*.py pair each
(pyutils/mask-faster-rcnn/lib/nets).  They differ in three places only:

  nfilt : 1 = one dynamic filter `dynamic_fc`                     (network.py:475-479)
          7 = `dynamic_fc_0..6` on spatial masks + `response_fc`  (network_7f.py:475-534)
  gate  : 'linear'  net_conv * response                           (NET:562)
          'sigmoid' net_conv * sigmoid(response) + response BCE   (network_7f_response.py:411-419,543-545;
                                                                    network_cycle_response.py:415-423,568-570)
  cap   : None | 'mask' (layer4 on the map, all + GT-masked pools, NET:415-440)
               | 'before_after' (layer4 on the map before and after the gating, network_cycle_response.py:425-439)
"""
VARIANTS = {
    'baseline': dict(nfilt=1, gate='linear', cap=None),            # nets/resnet_v1.py            train_baseline.sh
    'spatial': dict(nfilt=7, gate='linear', cap=None),             # nets/resnet_v1_7f.py         train_spatial.sh
    'response': dict(nfilt=7, gate='sigmoid', cap=None),           # nets/resnet_v1_7f_response.py train_response.sh
    'cycle': dict(nfilt=7, gate='linear', cap='mask'),             # nets/resnet_v1_cycle_res5_2.py train_cycle.sh
    'cycle_response': dict(nfilt=7, gate='sigmoid', cap='before_after'),   # nets/resnet_v1_cycle_response.py train_cycle_response.sh
    # VGG16 / Faster R-CNN variant (nets/vgg16.py + nets/network_vgg.py, train_vgg.sh): conv5_3 map, 14x14 crop + 2x2 max pool,
    # fc6 / fc7, no mask branch, sigmoid gating + response loss
    'vgg': dict(nfilt=7, gate='sigmoid', cap=None, backbone='vgg', mask=False),
}
# The solver of each variant (the reference has one model/train_val*.py per entry point, tools/train*.py:22-24).  Their construct_graph()
# differ in two things:
#   lang_lr_mult : learning-rate factor of every parameter whose key contains 'rnn_encoder', 'dynamic_fc' or 'response' - x10 in
#                  train_val.py:193-198 (baseline AND spatial), train_val_response.py:193-198, train_val_vgg.py:193-198; the same lines are
#                  commented out in train_val_cycle.py:199-204 and train_val_cycle_response.py:193-198 (factor 1)
#   config       : the config module the solver imports - model/config_vgg.py for VGG (train_val_vgg.py:12: WEIGHT_DECAY 5e-4, DOUBLE_BIAS True,
#                  config_vgg.py:28,40), model/config.py for the rest
SOLVERS = {
    'baseline': dict(module='train_val', lang_lr_mult=10.0, config='config'),
    'spatial': dict(module='train_val', lang_lr_mult=10.0, config='config'),
    'response': dict(module='train_val_response', lang_lr_mult=10.0, config='config'),
    'cycle': dict(module='train_val_cycle', lang_lr_mult=1.0, config='config'),
    'cycle_response': dict(module='train_val_cycle_response', lang_lr_mult=1.0, config='config'),
    'vgg': dict(module='train_val_vgg', lang_lr_mult=10.0, config='config_vgg'),
}
LANG_LR_KEYS = ('rnn_encoder', 'dynamic_fc', 'response')


def solver_cfg(variant):
    """the `cfg` object the variant's solver reads its TRAIN.* hyper-parameters from"""
    import importlib
    return importlib.import_module('lang2seg_amd.model.' + SOLVERS[variant]['config']).cfg


_ALL = ['rpn_cross_entropy', 'rpn_loss_box', 'cross_entropy', 'loss_box', 'loss_mask', 'loss_response', 'loss_caption', 'total_loss']
# slot of each loss in the device loss[8] buffer (include/lang2seg_hip.h L2S_LOSS_*)
SLOT = dict(rpn_cross_entropy=0, rpn_loss_box=1, cross_entropy=2, loss_box=3, loss_mask=4, loss_caption=5, total_loss=6, loss_response=7)


def loss_names(variant):
    """order of the floats `train_step` returns in that variant (NET:702-719 and its siblings)."""
    v = VARIANTS[variant]
    return [k for k in _ALL if not (k == 'loss_response' and v['gate'] != 'sigmoid') and not (k == 'loss_caption' and v['cap'] is None)
            and not (k == 'loss_mask' and not v.get('mask', True))]
