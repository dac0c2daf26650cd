"""resnetv1 of the reference's nets/resnet_v1.py + nets/network.py (train_baseline.sh): one dynamic filter, 6 losses.
Same constructor / create_architecture / train_step as every variant; see nets/variants.py and nets/resnet_v1.py."""
from .resnet_v1 import resnetv1 as _Base


class resnetv1(_Base):
    variant = 'baseline'
