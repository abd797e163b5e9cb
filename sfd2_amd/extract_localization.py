"""Host driver pieces mirroring extract_localization.py (reference): the named confs
(:25-120), get_model (:208-218) and the per-image post-processing of main (:245-272).
Image decoding / cubic resize (cv2) and HDF5 writing (h5py) are outside the hot path
(SURVEY.md section 8f rank 1); features are returned as the dict the reference writes."""
import os.path as osp

import numpy as np

from .extractor import extract_resnet_return
from .model import ResSegNetV2

_W = "weights/20220810_ressegnetv2_wapv2_ce_sd2mfsf_uspg.pth"


def _conf(n, r):
    name = f"ressegnetv2-20220810-wapv2-sd2mfsf-uspg-0001-n{n}-r{r}"
    return name, {
        'output': 'feats-' + name,
        'model': {'name': 'ressegnetv2', 'use_stability': True, 'max_keypoints': n, 'conf_th': 0.001,
                  'multiscale': False, 'scales': [1.0], 'model_fn': _W},
        'preprocessing': {'grayscale': False, 'resize_max': r},
        'mask': False,
    }


# the ressegnetv2 entries of extract_localization.py:25-120
confs = dict(_conf(n, r) for n, r in ((4096, 1600), (3000, 1600), (2000, 1600), (4096, 1024), (3000, 1024), (2000, 1024)))


def get_model(model_name, weight_path=None, use_stability=False, state_dict=None, device=0):
    """extract_localization.py:208-218.  state_dict may be given directly (numpy / torch dict)
    when the checkpoint is not a file; otherwise weight_path is read with torch.load."""
    if model_name != 'ressegnetv2':
        raise NotImplementedError("only 'ressegnetv2' is on the hot path (SURVEY.md section 2 #1)")
    model = ResSegNetV2(outdim=128, require_stability=use_stability).eval()
    if state_dict is None:
        import torch
        if not osp.exists(weight_path):
            raise FileNotFoundError(weight_path)
        state_dict = torch.load(weight_path, map_location='cpu')['model']
    model.load_state_dict(state_dict, strict=False)
    model.cuda(device)
    return model, extract_resnet_return


def rescale_keypoints(keypoints, original_size, size):
    """extract_localization.py:258-263: kp = (kp + .5) * (orig / size) - .5 with float32 scales."""
    scales = (np.asarray(original_size) / np.asarray(size)).astype(np.float32)
    return (keypoints + .5) * scales[None] - .5


def extract_one(model, extractor, image, original_size, conf):
    """One iteration of main's loop (extract_localization.py:245-272): image [1,3,H,W] in [0,1].
    Returns the group the reference writes: keypoints (N,2) f64, descriptors (128,N) f64,
    scores (N,) f64, image_size (2,) int."""
    pred = extractor(model, img=image, topK=conf["model"]["max_keypoints"], mask=None,
                     conf_th=conf["model"]["conf_th"], scales=conf["model"]["scales"])
    pred['descriptors'] = pred['descriptors'].transpose()
    pred['image_size'] = np.asarray(original_size)
    size = np.array(image.shape[-2:][::-1])
    pred['keypoints'] = rescale_keypoints(pred['keypoints'], original_size, size)
    return pred


def main(conf, images, export_dir, state_dict=None, device=0, tag=None):
    """extract_localization.py:221-275 without the image decoder: ``images`` yields
    {'name': str, 'image': uint8 [H,W,3] RGB (or float [3,H,W] in [0,1]), 'original_size': (w, h)}
    (what ImageDataset.__getitem__ returns before / after its astype-and-divide, :157-190); one
    group per image goes to the feature store with the reference's dataset names and dtypes.
    uint8 images are converted on the device.  Returns the store path."""
    import os
    from .feature_io import open_store, write_features
    model, extractor = get_model(conf['model']['name'], weight_path=conf['model']['model_fn'],
                                 use_stability=conf['model']['use_stability'], state_dict=state_dict, device=device)
    os.makedirs(str(export_dir), exist_ok=True)
    path = os.path.join(str(export_dir), conf['output'] + '.h5')
    store = open_store(path, 'a')
    try:
        for data in images:
            if tag is not None and data['name'].find(tag) < 0:
                continue
            img = data['image']
            if img.dtype == np.uint8:
                size = np.array(img.shape[:2][::-1])
                feed = img
            else:
                size = np.array(img.shape[-2:][::-1])
                feed = img[None] if img.ndim == 3 else img
            pred = extractor(model, img=feed, topK=conf["model"]["max_keypoints"], mask=None,
                             conf_th=conf["model"]["conf_th"], scales=conf["model"]["scales"])
            pred['descriptors'] = pred['descriptors'].transpose()
            pred['image_size'] = original_size = np.asarray(data['original_size'])
            pred['keypoints'] = rescale_keypoints(pred['keypoints'], original_size, size)
            write_features(store, data['name'], pred)
    finally:
        store.close()
    return getattr(store, 'path', path)
