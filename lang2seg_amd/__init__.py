"""lang2seg_amd — MI355X-native (gfx950) implementation of the lang2seg joint
forward/backward training hot path.  Host side mirrors the reference's Python
interface (nets.resnet_v1_cycle_res5_2.resnetv1, model.train_val_cycle.train_net,
loaders' blobs contract); all arithmetic runs in the hand-written HIP kernels of
lang2seg_amd/csrc behind the C ABI declared in include/lang2seg_hip.h."""
__version__ = '0.1.0'
