"""sfd2_amd: MI355X-native SFD2 feature extraction + matching hot path.

Host side (Python) mirrors the reference's operator API:
    sfd2_amd.model.ResSegNetV2                     <- nets/sfd2.py:259 ResSegNetV2 (.det)
    sfd2_amd.extractor.extract_resnet_return       <- nets/extractor.py:97
    sfd2_amd.matchers.nearest_neighbor.NearestNeighbor <- hloc/matchers/nearest_neighbor.py:27
    sfd2_amd.matcher.Matcher                       <- it_loc/matcher.py:85
All device work lives in libsfd2hip.so (C-ABI: include/sfd2_hip.h).
"""
__version__ = "0.1.0"
