"""resnetv1 of the reference's nets/resnet_v1_7f_response.py + nets/network_7f_response.py (train_response.sh): sigmoid gating + response loss, 7 losses.
Same constructor / create_architecture / train_step as every variant; see nets/variants.py and nets/resnet_v1.py."""
from .resnet_v1 import resnetv1 as _Base


class resnetv1(_Base):
    variant = 'response'
