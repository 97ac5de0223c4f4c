#!/usr/bin/env python3
"""Label the Gaussians of a 3DGS PLY by majority vote over segmented camera views — MI355X build.

Drop-in for the reference script of the same name: same functions, same CLI flags
(--ply_file --camera_file --input_dir --output_dir --output_file --model), same PLY-in / PLY-out
contract.  The per-Gaussian projection + vote (reference lines 43-82, 252-308) runs in hand-written HIP
kernels behind libgsx.so; PLY parsing/writing is native too.  The 2-D segmentation networks stay what
they are in the reference (Hugging Face / Ultralytics models, imported lazily): they need weights that
are fetched by name, so `--segmap_dir` lets a run consume the `<image>_segmap.npy` files the reference
itself writes next to every segmented image (reference line 165) instead of running a network.

Added flags: --segmap_dir DIR, --n_classes C (default: 150 for the ADE20K models, 80 for YOLO), --gpus N
(informational).  Several GPUs: `python -m torch.distributed.run --nproc-per-node N deep_learning_segmentation.py
...` — every rank segments and stages its contiguous block of the cameras, the packed class maps are all-gathered over
RCCL and every rank votes its slab of the Gaussians (3d_gaussian_splatting_project_amd/dist.py, protocol v4).
"""
import argparse
import importlib
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
gsx = importlib.import_module("3d_gaussian_splatting_project_amd")
from importlib import import_module as _imp

_ply = _imp("3d_gaussian_splatting_project_amd.ply_io")

N_CLASSES = {"segformer": 150, "mask2former": 150, "yolo": 80}


def load_cameras(camera_file):
    """cameras.json -> list of camera dicts."""
    return gsx.load_cameras(camera_file)


def load_gaussians(ply_file):
    """Returns (gaussians, plydata): a structured array whose 'position' field is filled (scale and
    rotation stay zero, exactly like the reference) and the opened PLY for save_labeled_ply."""
    plydata = _ply.PlyData.read(ply_file)
    vertices = plydata["vertex"]
    gaussians = np.zeros(len(vertices), dtype=[("position", np.float32, 3), ("scale", np.float32, 3), ("rotation", np.float32, 4)])
    for axis, name in enumerate("xyz"):
        gaussians["position"][:, axis] = vertices.column_f32(name)
    return gaussians, plydata


def project_gaussian(position, camera):
    """(x, y) pixel of a Gaussian centre in `camera`, or None — evaluated by the device kernel."""
    return gsx.project_gaussian(position, camera)


def initialize_model(model_type, device):
    """Load one of the reference's three segmentation networks (needs their weights to be available)."""
    if model_type == "segformer":
        from transformers import SegformerForSemanticSegmentation, SegformerImageProcessor
        name = "nvidia/segformer-b5-finetuned-ade-640-640"
        return SegformerImageProcessor.from_pretrained(name), SegformerForSemanticSegmentation.from_pretrained(name).to(device)
    if model_type == "mask2former":
        from transformers import AutoImageProcessor, Mask2FormerForUniversalSegmentation
        name = "facebook/mask2former-swin-large-ade-semantic"
        return AutoImageProcessor.from_pretrained(name), Mask2FormerForUniversalSegmentation.from_pretrained(name).to(device)
    if model_type == "yolo":
        from ultralytics import YOLO
        return None, YOLO("yolo11x-seg.pt")
    raise ValueError(f"Unknown model type: {model_type}")


def segment_image(image_path, output_dir, processor, model, device, model_type):
    """int32 (H, W) class map of one image, -1 = no class; also saved as <image>_segmap.npy.
    OUT OF SCOPE of this build: kept minimal so that the CLI still runs end to end where the
    networks' weights are available; it cannot be tested here (no weights, no network).  The hot path only depends on its
    OUTPUT contract (a 2-D integer map), which `--segmap_dir` feeds directly.  Nearest-neighbour resizes use
    torch's legacy 'nearest' (source index = floor(dst * scale)), the rule of the reference's cv2.INTER_NEAREST."""
    import torch
    from PIL import Image
    os.makedirs(output_dir, exist_ok=True)
    image = Image.open(image_path)
    width, height = image.size
    if model_type == "yolo":
        res = model(image, verbose=False)[0]
        seg_map = np.full(res.orig_shape, -1, dtype=np.int32)
        if getattr(res, "masks", None) is not None and res.boxes is not None:
            masks = torch.nn.functional.interpolate(res.masks.data[None].float(), size=tuple(res.orig_shape), mode="nearest")[0]
            for mask, box in zip(masks, res.boxes):
                if float(box.conf[0]) > 0.5:
                    seg_map = np.where(mask.cpu().numpy() > 0.5, int(box.cls[0]), seg_map)
        seg_map = seg_map.astype(np.int32)
    else:
        inputs = {k: v.to(device) for k, v in processor(images=image, return_tensors="pt").items()}
        with torch.no_grad():
            outputs = model(**inputs)
        if model_type == "segformer":
            low = outputs.logits.argmax(dim=1)[None].float()
            seg_map = torch.nn.functional.interpolate(low, size=(height, width), mode="nearest")[0, 0].cpu().numpy().astype(np.int32)
        else:
            result = processor.post_process_instance_segmentation(outputs, target_sizes=[(height, width)])[0]
            seg_map = result["segmentation"].cpu().numpy().astype(np.int32)
    base = os.path.splitext(os.path.basename(image_path))[0]
    np.save(os.path.join(output_dir, f"{base}_segmap.npy"), seg_map)
    return seg_map


def _image_size(path):
    from PIL import Image
    with Image.open(path) as im:
        return im.size


def assign_labels(gaussians, cameras, input_dir, output_dir, model_type="mask2former", segmap_dir=None, n_classes=None,
                  ctx=None):
    """Majority-vote label per Gaussian (int32, -1 = never visible), bit-identical to the reference.

    Cameras whose `<img_name>.png` is missing from input_dir are skipped with a warning, as in the
    reference; the remaining ones are voted in order (the order decides ties)."""
    if model_type not in N_CLASSES:
        raise ValueError(f"Unknown model type: {model_type}")
    n_classes = n_classes or N_CLASSES[model_type]
    todo = []
    for camera in cameras:
        img_path = os.path.join(input_dir, camera["img_name"] + ".png")
        if not os.path.exists(img_path):
            print(f"Warning: Image {camera['img_name']} not found")
            continue
        todo.append((camera, img_path))
    processor = model = device = None
    if segmap_dir is None and todo:
        import torch
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        processor, model = initialize_model(model_type, device)
    world, rank = _world()
    own = ctx is None
    ctx = ctx or gsx.Context()
    if own:
        gsx.bind_to_gpu_numa_node(ctx.device)   # maps, pinned staging and packer threads next to this rank's GPU
    try:
        # one process per GPU: this rank segments and stages its contiguous block of the processed cameras; the packed
        # maps are then all-gathered and every rank votes its slab of the Gaussians (dist.py, protocol v4).  A rank
        # may end up without any camera (more GPUs than images): it still takes part in every collective.
        lo, hi = gsx.dist.view_range(len(todo), rank, world) if world > 1 else (0, len(todo))
        total = max(1, len(todo))
        ctx.upload_positions(gaussians)
        ctx.vote_begin(n_classes, min(lo, total - 1), total)
        for camera, img_path in todo[lo:hi]:
            print(f"Processing image {os.path.basename(img_path)}...")
            if segmap_dir is not None:
                seg_map = np.load(os.path.join(segmap_dir, camera["img_name"] + "_segmap.npy"))
            else:
                seg_map = segment_image(img_path, output_dir, processor, model, device, model_type)
            ctx.vote_view(camera, seg_map, _image_size(img_path))
        if world > 1:
            return gsx.dist.exchange_labels_gather(gsx.dist.GpuGatherShard(ctx), cap_views=total)
        return ctx.vote_finalize()
    finally:
        if own:
            ctx.close()


def _shutdown(failed=False):
    """Leave a torch.distributed run.  After a failure on THIS rank the peers may be blocked in a collective this rank
    will never join: no barrier then, the process ends with a non-zero code and the launcher takes the others down."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            if failed:
                import sys
                import traceback
                traceback.print_exc()
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(1)
            dist.barrier()
            dist.destroy_process_group()


def _world():
    """(world size, rank) of a torch.distributed run (python -m torch.distributed.run ... this_script.py);
    initialises the process group on first use.  (1, 0) for a plain single-process run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 1, 0
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GSX_DIST_BACKEND", "nccl")      # "gloo": functional rehearsal on one GPU
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return dist.get_world_size(), dist.get_rank()


def save_labeled_ply(output_file, plydata, labels):
    """All original vertex properties + a trailing `int label`, binary little endian."""
    plydata.write(output_file, labels=np.asarray(labels, dtype=np.int32))


def main(argv=None):
    parser = argparse.ArgumentParser(description="Add labels to gaussians PLY file")
    parser.add_argument("--ply_file", help="Input PLY file with gaussians data")
    parser.add_argument("--camera_file", help="JSON file with camera data")
    parser.add_argument("--input_dir", help="Directory containing input images")
    parser.add_argument("--output_dir", help="Output directory to saved segmented input images")
    parser.add_argument("--output_file", help="Output PLY file with labels")
    parser.add_argument("--model", choices=["segformer", "mask2former", "yolo"], default="mask2former",
                        help="Choose segmentation model: mask2former or yolo")
    parser.add_argument("--segmap_dir", default=None, help="read <img_name>_segmap.npy from here instead of running a model")
    parser.add_argument("--n_classes", type=int, default=None, help="number of classes (labels -1..C-1)")
    parser.add_argument("--gpus", type=int, default=1, help="GPUs of this node (N>1: launch with torch.distributed.run)")
    args = parser.parse_args(argv)

    print("Loading cameras...")
    cameras = load_cameras(args.camera_file)
    print("Loading gaussians...")
    gaussians, plydata = load_gaussians(args.ply_file)
    print("Assigning labels...")
    try:
        labels = assign_labels(gaussians, cameras, args.input_dir, args.output_dir, model_type=args.model,
                               segmap_dir=args.segmap_dir, n_classes=args.n_classes)
    except BaseException:
        _shutdown(failed=True)
        raise
    _shutdown()
    if int(os.environ.get("RANK", "0")) != 0:
        return          # every rank holds the same labels; rank 0 writes the file
    print("Saving labeled PLY file...")
    save_labeled_ply(args.output_file, plydata, labels)
    print(f"Done! Labeled PLY file saved as {args.output_file}")

    values, counts = np.unique(labels, return_counts=True)
    print("\nLabel statistics:")
    print(f"Total gaussians: {len(labels)}")
    print(f"Number of unique labels: {len(values)}")
    print("Label counts:")
    for value, count in zip(values, counts):
        print(f"Label {value}: {count} gaussians ({100 * count / len(labels):.2f}%)")


if __name__ == "__main__":
    main()
