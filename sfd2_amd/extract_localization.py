"""Host driver mirroring extract_localization.py (reference): the named confs (:25-120), ImageDataset (:122-202),
get_model (:208-218) and main (:221-279).  The decoder is whatever the host has (PIL here; the reference uses
cv2.imread); everything after the decoder -- astype(float32), the cubic resize_max downsize, / 255, the network,
selection and descriptors -- runs on the MI355X.  Work lists shard round-robin over `world` processes (one per GPU,
no data-path collective); rank 0 merges the per-rank parts in index order."""
import os
import os.path as osp
import threading

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


def get_model(model_name, weight_path=None, use_stability=False, state_dict=None, device=0, precision="f16x3"):
    """extract_localization.py:208-218.  state_dict may be given directly (numpy / torch dict)
    when the checkpoint is not a file; otherwise weight_path is read with torch.load (the reference's
    {'model': state_dict, ...} checkpoint layout, :213-215).  precision: see ResSegNetV2 -- the drop-in default
    is 'f16x3' (the strict mode's tolerances on the fp16 matrix path: descriptors 2e-5); 'f16x3d' is north_star's contract as written (the
    strict mode's key points -- 'f16x3''s, bit for bit -- and descriptors within 1e-3; 1.12x the speed of 'f16x3'), 'f16c' the
    tolerance-conformant throughput mode (descriptors 1e-3, key-point SET IoU >= 0.985), 'f16' an approximation."""
    if model_name != 'ressegnetv2':
        raise NotImplementedError("only 'ressegnetv2' is on the hot path (SURVEY.md section 2 #1)")
    model = ResSegNetV2(outdim=128, require_stability=use_stability, precision=precision).eval()
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


def resized_shape(w, h, resize_max=None, resize_force=False):
    """extract_localization.py:172-175 -> (w_new, h_new), or (w, h) when the image is left alone."""
    if resize_max and (resize_force or max(w, h) > resize_max):
        scale = resize_max / max(h, w)
        return int(round(w * scale)), int(round(h * scale))
    return w, h


_RGBX_OK = None      # does PIL let us paste into an RGBX view of our own buffer?  (checked once against the repacking path)


def _paste_rgbx(im, reserve):
    """PIL keeps an RGB image as four bytes per pixel; np.asarray(im) repacks it to three under the interpreter lock (~1.6 ms per 1600x1200
    image: at 16 decoder threads that lock is what bounds the pool, 750-850 images/s).  Image.paste into an RGBX image that VIEWS the caller's
    buffer copies the rows with the lock released; the library takes the four-byte pixels as they are (SFD2_FLAG_IMG_U8_X).
    Returns uint8 [h, w, 4] in reserve's memory, or None when this PIL does not play along."""
    global _RGBX_OK
    if _RGBX_OK is False:
        return None
    from PIL import Image
    w, h = im.size
    try:
        out = reserve(h * w * 4).reshape(h, w, 4)
        view = Image.frombuffer("RGBX", (w, h), out, "raw", "RGBX", 0, 1)
        view.readonly = 0                      # (frombuffer marks a shared buffer read-only; paste would copy it away first)
        view.paste(im)
        if _RGBX_OK is None:                   # first image of the process: the same bytes as the repacking path?
            _RGBX_OK = bool(np.array_equal(out[:, :, :3], np.asarray(im, dtype=np.uint8)))
            if not _RGBX_OK:
                return None
        return out
    except Exception:
        _RGBX_OK = False
        return None


_CV2 = False         # not looked for yet


def _cv2():
    """The reference's decoder (cv2.imread, extract_localization.py:162-165) when it is importable; None -> PIL.  SFD2_DECODER = "pil" / "cv2" forces one
    ("cv2" raises when the import fails); default "auto".  Both sit on libjpeg(-turbo); where the two builds differ in their IDCT / chroma upsampling the
    pixels differ by a count or two, which is why the reference's own decoder is preferred wherever it exists."""
    global _CV2
    if _CV2 is False:
        want = os.environ.get("SFD2_DECODER", "auto").lower()
        _CV2 = None
        if want != "pil":
            try:
                import cv2
                _CV2 = cv2
            except Exception as e:
                if want == "cv2":
                    raise RuntimeError("SFD2_DECODER=cv2 but cv2 is not importable") from e
    return _CV2


def _read_rgb_u8_cv2(cv2, path, reserve, rgbx):
    """cv2.imread(path, IMREAD_COLOR) -> BGR; the channel reversal of :165 is a cvtColor straight into the caller's buffer (three- or four-byte pixels; cv2
    releases the interpreter lock for both calls)."""
    img = cv2.imread(str(path), cv2.IMREAD_COLOR)
    if img is None:
        raise ValueError(f'Cannot read image {str(path)}.')            # extract_localization.py:166-167
    h, w = img.shape[:2]
    if reserve is None:
        return np.ascontiguousarray(img[:, :, ::-1])
    if rgbx:
        out = reserve(h * w * 4).reshape(h, w, 4)
        res = cv2.cvtColor(img, cv2.COLOR_BGR2RGBA, dst=out)           # (the fourth byte is not read: SFD2_FLAG_IMG_U8_X)
    else:
        out = reserve(h * w * 3).reshape(h, w, 3)
        res = cv2.cvtColor(img, cv2.COLOR_BGR2RGB, dst=out)
    if res is not out and not np.shares_memory(res, out):              # (a cv2 that did not take dst)
        np.copyto(out, res)
    return out


def _read_rgb_u8(path, reserve=None, rgbx=False):
    """The decoder: uint8 [H,W,3] RGB (the reference: cv2.imread(..., IMREAD_COLOR)[:, :, ::-1], :162-165).
    reserve(nbytes) -> writable uint8 buffer: the pixels are put there (the pipelined driver's pinned buffers).
    rgbx (with reserve): [H,W,4] RGBX when PIL allows it (_paste_rgbx), else [H,W,3]."""
    cv2 = _cv2()
    if cv2 is not None:
        return _read_rgb_u8_cv2(cv2, path, reserve, rgbx)
    try:
        from PIL import Image
    except Exception as e:  # pragma: no cover
        raise RuntimeError("no image decoder importable (cv2, PIL); pass decoded uint8 arrays instead of paths") from e
    try:
        with Image.open(path) as im:
            if im.mode != "RGB":
                im = im.convert("RGB")
            else:
                im.load()
            if reserve is None:
                return np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
            if rgbx:
                out = _paste_rgbx(im, reserve)
                if out is not None:
                    return out
            w, h = im.size
            out = reserve(h * w * 3).reshape(h, w, 3)
            np.copyto(out, np.asarray(im, dtype=np.uint8))
            return out
    except (OSError, ValueError) as e:
        raise ValueError(f'Cannot read image {str(path)}.') from e   # extract_localization.py:166-167


class ImageDataset:
    """extract_localization.py:122-202 without torch's Dataset base: same default_conf, the same file discovery
    (globs under root, or an image list), __getitem__ -> {'name', 'image', 'original_size'}.  'image' is the DECODED
    uint8 [H,W,3] RGB array; the float conversion and the cubic resize happen on the device (preprocess below), so
    'resize' carries the target (w, h) instead of the resized pixels.  grayscale is not on the SFD2 path."""
    default_conf = {
        'globs': ['*.jpg', '*.png', '*.jpeg', '*.JPG', '*.PNG'],
        'grayscale': False,
        'resize_max': None,
        'resize_force': False,
    }

    def __init__(self, root, conf, image_list=None, mask_root=None):
        from pathlib import Path
        self.conf = {**self.default_conf, **conf}
        if self.conf['grayscale']:
            raise NotImplementedError("grayscale input is not used by the ressegnetv2 confs")
        self.root = Path(root)
        self.paths = []
        if image_list is None:
            for g in self.conf['globs']:
                self.paths += list(self.root.glob('**/' + g))
            if len(self.paths) == 0:
                raise ValueError(f'Could not find any image in root: {root}.')
            self.paths = sorted(i.relative_to(self.root) for i in self.paths)
        else:
            with open(image_list, "r") as f:
                self.paths = [Path(l.strip()) for l in f.readlines() if l.strip()]
        self.mask_root = mask_root

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, idx):
        return self.load(idx)

    def load(self, idx, reserve=None, rgbx=False):
        """__getitem__ with the decoded pixels placed in reserve(nbytes) (see _read_rgb_u8); rgbx: four bytes per pixel when possible."""
        path = self.paths[idx]
        image = _read_rgb_u8(self.root / path, reserve, rgbx)
        h, w = image.shape[:2]
        return {'name': str(path), 'image': image, 'original_size': np.array((w, h)),
                'resize': resized_shape(w, h, self.conf['resize_max'], self.conf['resize_force'])}


def preprocess(model, image_u8, resize=None, bgr=False):
    """uint8 [H,W,3] -> torch cuda float32 [1,3,h,w] in [0,1]: ImageDataset.__getitem__'s astype / cv2.resize(INTER_CUBIC) /
    transpose / 255 (:168-186) on the device (sfd2_preprocess)."""
    import torch
    from . import _lib
    ctx = model.context
    H, W = image_u8.shape[:2]
    w_new, h_new = resize if resize is not None else (W, H)
    out = torch.empty((1, 3, h_new, w_new), dtype=torch.float32, device=torch.device("cuda", ctx.device))
    a = np.ascontiguousarray(image_u8, dtype=np.uint8)
    _lib.check(ctx.lib.sfd2_preprocess(ctx.h, a.ctypes.data, 0, H, W, _lib.FLAG_IMG_BGR if bgr else 0, h_new, w_new,
                                       out.data_ptr()))
    return out


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


def _part_path(export_dir, conf, rank, world):
    base = os.path.join(str(export_dir), conf['output'])
    return base + '.h5' if world == 1 else f'{base}.part{rank}of{world}.h5'


def _extract_pipelined(model, extractor, conf, images, indices, tag, store, names, num_workers, writers, depth, lanes=1):
    """The loop of main() with its three stages running side by side (sfd2_amd/pipeline.py): `num_workers` decoder threads
    (the reference: DataLoader(num_workers=4), extract_localization.py:230-233) fill pinned buffers in item order, the
    device runs up to `depth` asynchronous extractions, `writers` threads build the float64 groups and append them.
    Single-scale uint8 items on a HIP model go through the asynchronous C-ABI; anything else (float images, a scale
    pyramid, a stub extractor in the CPU tests) calls `extractor` synchronously between the same decode and writer
    stages.  Groups are written in item order per writer, with the arithmetic of the serial loop."""
    from .feature_io import write_features
    from .pipeline import AsyncExtractor, OrderedPrefetch, PinnedPool, WriterPool, slot_arrays
    mconf = conf["model"]
    top_k, conf_th, scales = mconf["max_keypoints"], mconf["conf_th"], [float(x) for x in mconf["scales"]]
    use_async = (extractor is extract_resnet_return and getattr(model, "context", None) is not None
                 and scales == [1.0] and top_k > 0)
    ax = AsyncExtractor(model, top_k, conf_th, depth=depth, slots=depth + 2 * writers + 2, lanes=lanes) if use_async else None
    pool = PinnedPool(num_workers + depth + 2) if use_async else None
    lock = threading.Lock()

    def load(idx, buf):
        if hasattr(images, "load"):
            data = images.load(idx, buf.reserve, rgbx=True) if buf is not None else images.load(idx, None)
        else:
            data = images[idx]
            img = data["image"]
            if buf is not None and getattr(img, "dtype", None) == np.uint8:
                pinned = buf.reserve(img.size).reshape(img.shape)
                np.copyto(pinned, img)
                data = dict(data, image=pinned)
        return idx, data, buf

    def write(job):
        idx, name, original_size, size, arrays, slot = job
        try:
            if slot is not None:
                arrays = slot_arrays(slot)
            kp, sc, de = arrays
            pred = {'keypoints': rescale_keypoints(kp, original_size, size), 'descriptors': de.transpose(), 'scores': sc,
                    'image_size': original_size}
            write_features(store, name, pred)
        finally:
            if slot is not None:
                ax.release(slot)        # (the descriptors are a VIEW of the slot's pinned block until the store has copied them)
        with lock:
            names.append((idx, name))

    if tag is not None and hasattr(images, "paths"):
        indices = [i for i in indices if str(images.paths[i]).find(tag) >= 0]
    pf = OrderedPrefetch(load, indices, num_workers, window=num_workers + 2,
                         claim=(lambda: pool.acquire(block=False)) if pool is not None else None)
    def drop(job):                  # a job queued behind a failed write: its slot still goes back (the producer may be waiting for one)
        if job[5] is not None:
            ax.release(job[5])

    wp = WriterPool(write, workers=writers, maxsize=4 * writers, on_drop=drop)

    def drain_one():
        slot = ax.finish()
        if slot.inbuf is not None:
            pool.release(slot.inbuf)
            slot.inbuf = None
        idx, name, original_size = slot.meta
        wp.put((idx, name, original_size, np.array(slot.size), None, slot))

    try:
        while True:
            pf.top_up()
            if pf.ready and (ax is None or len(ax.inflight) < depth):
                idx, data, buf = pf.pop()
                if tag is not None and data['name'].find(tag) < 0:
                    if buf is not None:
                        pool.release(buf)
                    continue
                img = data['image']
                original_size = np.asarray(data['original_size'])
                if ax is not None and getattr(img, "dtype", None) == np.uint8 and img.ndim == 3:
                    H, W = img.shape[:2]
                    ax.submit(img, H, W, tuple(data.get('resize', (W, H))), (idx, data['name'], original_size), inbuf=buf)
                    continue
                if ax is not None:
                    while ax.inflight:          # keep the store in item order: everything queued before this item first
                        drain_one()
                if buf is not None:
                    pool.release(buf)
                feed, size = _feed_of(model, data)
                pred = extractor(model, img=feed, topK=top_k, mask=None, conf_th=conf_th, scales=mconf["scales"])
                wp.put((idx, data['name'], original_size, size, (pred['keypoints'], pred['scores'], pred['descriptors']), None))
            elif ax is not None and ax.inflight:
                drain_one()
            elif pf.exhausted:
                break
    finally:
        pf.close()
        wp.close()
    # the serial loop appends in item order; writers finish out of order by a few items
    names.sort()


def _feed_of(model, data):
    """What main()'s loop hands to the extractor for one item, and the size the key points refer to."""
    img = data['image']
    if img.dtype == np.uint8:
        H, W = img.shape[:2]
        resize = tuple(data.get('resize', (W, H)))
        if resize != (W, H):
            return preprocess(model, img, resize), np.array(resize)      # device: float, cubic resize, / 255
        return img, np.array((W, H))                                     # converted while conv1a stages its patch
    return (img[None] if img.ndim == 3 else img), np.array(img.shape[-2:][::-1])


def main(conf, images, export_dir, state_dict=None, device=0, tag=None, precision="f16x3", world=1, rank=0,
         barrier=None, model_and_extractor=None, num_workers=4, writers=2, depth=None, lanes=None, affinity=None):
    """extract_localization.py:221-279.  ``images``: an ImageDataset (decoded from files, resized per
    conf['preprocessing']) or any indexable / iterable of {'name', 'image': uint8 [H,W,3] RGB or float [3,H,W] in
    [0,1], 'original_size': (w, h)[, 'resize': (w, h)]}.  uint8 input is exact only together with the device resize
    or for images that need none: the reference resizes the FLOAT image (:168-177), which a caller-side uint8 resize
    cannot reproduce.  One group per image goes to the feature store with the reference's dataset names and dtypes.

    Multi-GPU: with world > 1 this process takes items rank, rank + world, ... (extract_localization.py:240
    is the loop that shards), writes them to its own part store and, after ``barrier()`` (torch.distributed.barrier
    or equivalent; one process per GPU), rank 0 merges the parts into the final store in item order.

    num_workers > 0 (default 4, the reference's DataLoader(num_workers=4), extract_localization.py:230-233): the pipelined
    loop (_extract_pipelined: that many decoder threads, `depth` images in flight on the device (default: three per lane) over `lanes`
    contexts (HIP streams; default: 2 from 64 images on, else 1 -- the second context is made once per model, model.lanes), `writers` writer
    threads); 0: the reference's loop body strictly in turn per image.  Both write the same groups.
    affinity: None (default) = with world > 1 the process pins itself, before its decoder / writer threads exist, to its share of the CPUs of the socket
    GPU `device` hangs off (sharding.pin_to_gpu_socket; SFD2_CPU_AFFINITY=0 disables); True / False force it on / off.
    Returns the final store path (rank 0) or the part path (other ranks)."""
    from .feature_io import open_store, write_features
    from .sharding import pin_to_gpu_socket, shard_indices
    if affinity or (affinity is None and world > 1):
        pin_to_gpu_socket(device, local_world=world if world > 1 else None, enable=True if affinity else None)
    if model_and_extractor is None:
        model, extractor = get_model(conf['model']['name'], weight_path=conf['model']['model_fn'],
                                     use_stability=conf['model']['use_stability'], state_dict=state_dict, device=device,
                                     precision=precision)
    else:
        model, extractor = model_and_extractor
    os.makedirs(str(export_dir), exist_ok=True)
    if not hasattr(images, '__getitem__'):
        images = list(images)
    n_items = len(images)
    path = _part_path(export_dir, conf, rank, world)
    store = open_store(path, 'a' if world == 1 else 'w')
    names = []
    try:
        if num_workers and num_workers > 0:
            # two contexts (HIP streams) taking the images in turn: +7-9 % on the device (tile tails and small launches of one image filled by the
            # other's kernels); the second context costs a weight upload once per model (model.lanes), so short jobs stay on one
            idx_list = list(shard_indices(n_items, rank, world))
            n_lanes = max(1, int(lanes)) if lanes is not None else (2 if len(idx_list) >= 64 else 1)
            n_depth = max(1, int(depth)) if depth is not None else 3 * n_lanes
            _extract_pipelined(model, extractor, conf, images, idx_list, tag, store, names,
                               int(num_workers), max(1, int(writers)), n_depth, n_lanes)
        for idx in (() if num_workers and num_workers > 0 else shard_indices(n_items, rank, world)):
            data = images[idx]
            if tag is not None and data['name'].find(tag) < 0:
                continue
            feed, size = _feed_of(model, data)
            pred = extractor(model, img=feed, topK=conf["model"]["max_keypoints"], mask=None,
                             conf_th=conf["model"]["conf_th"], scales=conf["model"]["scales"])
            pred['descriptors'] = pred['descriptors'].transpose()
            pred['image_size'] = original_size = np.asarray(data['original_size'])
            pred['keypoints'] = rescale_keypoints(pred['keypoints'], original_size, size)
            write_features(store, data['name'], pred)
            names.append((idx, data['name']))
        actual = getattr(store, 'path', getattr(store, 'filename', path))
    finally:
        store.close()
    if world == 1:
        return actual
    import json
    with open(path + '.index.json', 'w') as f:      # (item index, group name) in this rank's write order
        json.dump(names, f)
    if barrier is not None:
        barrier()
    if rank != 0:
        return actual
    return merge_parts(export_dir, conf, world)


def merge_parts(export_dir, conf, world):
    """Rank 0, after every rank has closed its part: copy the groups of the `world` part stores into the final
    store in item order (each part carries an index of (item index, group name))."""
    import json
    from .feature_io import open_store, write_features
    items = []
    for r in range(world):
        with open(_part_path(export_dir, conf, r, world) + '.index.json') as f:
            items += [(int(i), name, r) for i, name in json.load(f)]
    items.sort()
    final = open_store(_part_path(export_dir, conf, 0, 1), 'a')
    parts = [open_store(_part_path(export_dir, conf, r, world), 'r') for r in range(world)]
    try:
        for _, name, r in items:
            if name in final:
                continue
            g = parts[r][name]
            write_features(final, name, {k: np.asarray(g[k].__array__()) for k in g.keys()})
        actual = getattr(final, 'path', getattr(final, 'filename', None))
    finally:
        final.close()
        for p in parts:
            p.close()
    return actual


def _dist_env():
    """(world, rank, local_rank, barrier) of a `python -m torch.distributed.run --nproc-per-node N -m sfd2_amd....` launch (one process per GPU), or a
    single process.  The drivers have no data-path collective: the process group (gloo: a barrier is all it carries) only lines the ranks up before rank 0
    merges their part stores."""
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1:
        return 1, 0, local, None
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    return world, rank, local, dist.barrier


def cli(argv=None):
    """The reference script's command line (extract_localization.py:281-291: --image_dir, --image_list, --tag, --mask_dir, --export_dir, --conf) plus what this
    path adds (--weights overriding the conf's model_fn, --precision, --num_workers, --device).  Under torchrun every rank takes its share of the images on
    its own GPU (LOCAL_RANK) and rank 0 merges."""
    import argparse
    from pathlib import Path
    ap = argparse.ArgumentParser(description="SFD2 feature extraction on MI355X (drop-in for the reference's extract_localization.py)")
    ap.add_argument('--image_dir', type=Path, required=True)
    ap.add_argument('--image_list', type=str, default=None)
    ap.add_argument('--tag', type=str, default=None)
    ap.add_argument('--mask_dir', type=Path, default=None, help="accepted for compatibility (the reference passes mask_root=None as well)")
    ap.add_argument('--export_dir', type=Path, required=True)
    ap.add_argument('--conf', type=str, default=next(iter(confs)), choices=list(confs.keys()))
    ap.add_argument('--weights', type=str, default=None, help="checkpoint file (default: the conf's model_fn, " + _W + ")")
    ap.add_argument('--precision', type=str, default="f16x3", choices=["f32", "f16x3", "f16x3d", "f16c", "f16"])
    ap.add_argument('--num_workers', type=int, default=4, help="decoder threads (the reference's DataLoader(num_workers=4)); 0 = the serial loop")
    ap.add_argument('--device', type=int, default=None, help="GPU index (default: LOCAL_RANK modulo the visible GPUs)")
    args = ap.parse_args(argv)
    conf = confs[args.conf]
    if args.weights:
        conf = {**conf, 'model': {**conf['model'], 'model_fn': args.weights}}
    world, rank, local, barrier = _dist_env()
    device = args.device
    if device is None:
        import torch
        device = local % max(1, torch.cuda.device_count())
    ds = ImageDataset(args.image_dir, conf['preprocessing'], image_list=args.image_list, mask_root=None)
    path = main(conf, ds, args.export_dir, device=device, tag=args.tag, precision=args.precision, world=world, rank=rank, barrier=barrier,
                num_workers=args.num_workers)
    if rank == 0:
        print(path)
    return path


if __name__ == '__main__':
    cli()
