"""Decode-ahead image batches for the evaluation / gallery-building loops (the caller side).

The reference feeds its models from a `DataLoader` (`dinov2salad/dinov2salad_validation.py:74`, `num_workers=0`;
`swin_transformer/val_and_test_swin_2.py:203-205`, 4 worker processes): decode + resize + normalise on the host, one
`.cuda()` per batch.  Here the resize / normalise run on the GPU (`preprocess.ResizeNormalize`, PIL-exact), so what is
left on the host is the image decode — ~1-2 ms per JPEG, i.e. 10x the GPU time of a 64-image step if done serially.
`ImageBatchLoader` keeps `depth` batches ahead of the consumer:

  * a thread pool decodes (PIL releases the GIL inside its codecs) straight into slots of a pinned host buffer.  What
    caps threads is the GIL-held part of a decode: PIL's raw export packs its 4-byte RGBX pixels to 3 under the lock
    (~0.4 ms per VGA frame).  With pyarrow present the pixel storage is handed over zero-copy (Arrow C data interface,
    8 us) and copied with one GIL-free memcpy as RGBX; the fourth byte is dropped on the GPU.  `scripts/loader_bench.py`,
    1536 VGA JPEGs, decode + copy: serial 0.64k images/s, 8 threads 2.5k (raw export) / 4.3k (Arrow hand-over).
    Worker PROCESSES (a `DataLoader` over the same plan) were measured and dropped: forking 8-15 workers from a process
    that holds the GPU context costs seconds; 0.4-1.4k images/s on the same list,
  * a finished batch goes to the device with one asynchronous copy on a dedicated copy stream,
  * the consumer's stream waits on that copy's event only — no host sync, the previous batch's kernels keep running.

Same bytes, same batches, same order as the serial `np.stack([np.asarray(Image.open(f).convert("RGB")) ...])` loop:
batches hold images of one size (mixed sizes are grouped per size, in order of first appearance) and carry the indices
of their rows in the file list.  On a CPU device the loader degrades to the same decode-ahead without pinning / streams
(used by the `-m "not gpu"` tests).
"""
from __future__ import annotations

import os
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from PIL import Image

try:
    import pyarrow as _pa
except ImportError:                                              # the raw export path needs nothing but PIL
    _pa = None


def batches_by_size(image_dir: str, filenames: Sequence[str], batch_size: int) -> List[Tuple[Tuple[int, int], List[int], List[str]]]:
    """[((W, H), row indices, filenames)] groups of equal image size, sizes in order of first appearance."""
    sizes = {}
    for i, f in enumerate(filenames):
        with Image.open(os.path.join(image_dir, f)) as im:      # header only
            sizes.setdefault(im.size, []).append(i)
    out = []
    for size, idxs in sizes.items():
        for lo in range(0, len(idxs), batch_size):
            sel = idxs[lo:lo + batch_size]
            out.append((size, sel, [filenames[i] for i in sel]))
    return out


def _decode_into(path: str, slot: np.ndarray) -> None:
    """slot [H,W,3]: PIL's raw export (one pass under the GIL that packs its 4-byte RGBX pixels to 3) + a copy.
    slot [H,W,4]: PIL's pixel storage itself, handed over zero-copy through the Arrow C data interface and copied with
    one GIL-free memcpy; the X byte is dropped on the GPU."""
    with Image.open(path) as im:
        rgb = im if im.mode == "RGB" else im.convert("RGB")                 # convert() of an RGB image is a plain copy
        if slot.shape[-1] == 4:
            rgb.load()
            try:
                arr = _pa.array(rgb).values.to_numpy(zero_copy_only=True).reshape(rgb.size[1], rgb.size[0], 4)
            except ValueError:                 # images above PIL's 16 MB block size live in several blocks: no zero-copy view
                arr = np.asarray(rgb)
                if arr.shape != slot.shape[:2] + (3,):
                    raise RuntimeError(f"{path}: decoded to {arr.shape}, its header promised {slot.shape[:2] + (3,)}")
                np.copyto(slot[..., :3], arr)  # the X byte of the slot is never read on the device
                return
        else:
            arr = np.asarray(rgb)
        if arr.shape != slot.shape:
            raise RuntimeError(f"{path}: decoded to {arr.shape}, its header promised {slot.shape}")
        np.copyto(slot, arr)


def _arrow_export_works() -> bool:
    """Pillow >= 11.2 exports an RGB image as fixed_size_list<uint8>[4] over its own storage; probe once."""
    if _pa is None:
        return False
    try:
        im = Image.new("RGB", (3, 2), (1, 2, 3))
        v = _pa.array(im).values.to_numpy(zero_copy_only=True).reshape(2, 3, 4)
        return bool((v[..., :3] == np.array([1, 2, 3], dtype=np.uint8)).all())
    except Exception:                                            # noqa: BLE001  any failure = the raw export path
        return False


class ImageBatchLoader:
    """Iterate `(row indices, filenames, uint8 [B,H,W,3] tensor on `device`)` with decode and H2D copy running ahead."""

    def __init__(self, image_dir: str, filenames: Sequence[str], batch_size: int, device, *,
                 workers: Optional[int] = None, depth: int = 3, export: str = "auto"):
        """export: how a decoded image leaves PIL — "raw" (np.asarray: a packing pass under the GIL), "arrow" (zero-copy
        RGBX hand-over + GIL-free memcpy, the fourth byte dropped on the device; needs pyarrow and Pillow >= 11.2) or
        "auto" (arrow on a CUDA device when it works)."""
        if export not in ("auto", "raw", "arrow"):
            raise ValueError("export must be 'auto', 'raw' or 'arrow'")
        self.image_dir, self.device = image_dir, torch.device(device)
        self.filenames = list(filenames)
        self.plan = batches_by_size(image_dir, self.filenames, batch_size)
        self.workers = workers if workers is not None else max(1, min(8, (os.cpu_count() or 2) - 1))   # more than 8 lose to the GIL
        self.depth = max(1, depth)
        self._cuda = self.device.type == "cuda"
        if export == "arrow" and not _arrow_export_works():
            raise RuntimeError("export='arrow' needs pyarrow and a Pillow with the Arrow C data interface")
        self.export = export if export != "auto" else ("arrow" if self._cuda and _arrow_export_works() else "raw")

    def __len__(self) -> int:
        return len(self.plan)

    def _host_buffer(self, ring: dict, shape) -> torch.Tensor:
        """Next host buffer of this shape from a ring of depth + 1 (pinned on CUDA): a buffer is reused only after the
        copy that read it has finished (its event is synchronised before the decode threads write again)."""
        slots = ring.setdefault(shape, {"bufs": [], "events": [], "next": 0})
        if len(slots["bufs"]) <= self.depth:
            slots["bufs"].append(torch.empty(shape, dtype=torch.uint8, pin_memory=self._cuda))
            slots["events"].append(None)
            i = len(slots["bufs"]) - 1
        else:
            i = slots["next"]
            slots["next"] = (i + 1) % len(slots["bufs"])
            if slots["events"][i] is not None:
                slots["events"][i].synchronize()
        slots["cur"] = i
        return slots["bufs"][i]

    def _to_device(self, host: torch.Tensor, copy_stream):
        """Asynchronous H2D copy of a pinned batch on the copy stream; the CONSUMER's stream waits for it."""
        with torch.cuda.stream(copy_stream):
            dev_t = torch.empty(host.shape, dtype=torch.uint8, device=self.device)
            dev_t.copy_(host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        dev_t.record_stream(cur)
        return dev_t, ev

    def __iter__(self) -> Iterator[Tuple[List[int], List[str], torch.Tensor]]:
        ring: dict = {}
        copy_stream = torch.cuda.Stream(device=self.device) if self._cuda else None
        pending = deque()
        plan = iter(self.plan)
        with ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="vpr-decode") as pool:
            def launch():
                item = next(plan, None)
                if item is None:
                    return False
                (W, H), idxs, names = item
                shape = (len(names), H, W, 4 if self.export == "arrow" else 3)
                buf = self._host_buffer(ring, shape)
                slot_of = ring[shape]["cur"]
                view = buf.numpy()
                futs = [pool.submit(_decode_into, os.path.join(self.image_dir, f), view[j]) for j, f in enumerate(names)]
                pending.append((idxs, names, buf, futs, shape, slot_of))
                return True

            for _ in range(self.depth):
                if not launch():
                    break
            while pending:
                idxs, names, buf, futs, shape, slot_of = pending.popleft()
                for f in futs:
                    f.result()                                   # re-raises a decode error here, in order
                if self._cuda:
                    dev_t, ev = self._to_device(buf, copy_stream)    # the consumer's stream, not the host, waits for the copy
                    ring[shape]["events"][slot_of] = ev
                else:
                    dev_t = buf.clone()
                if shape[-1] == 4:
                    dev_t = dev_t[..., :3].contiguous()          # RGBX -> RGB on the consumer's stream (after the copy's event)
                launch()                                         # keep `depth` batches decoding while the consumer computes
                yield idxs, names, dev_t
