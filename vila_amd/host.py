"""Host-side integer logic of the path (no GPU needed; covered by the `not gpu` tests):

  * `splice_plan`  — the index arithmetic behind `_embed` + `__batchify_sequence`
                     (llava/model/llava_arch.py:412-490, 528-555) as row maps for the gather kernels
  * `repack`       — `repack_multimodal_data`, non-SP branch (llava_arch.py:744-800) + `_get_unpad_data`
                     (llava/model/utils/packing.py:12-21): packed row, restarted positions, cu_seqlens
  * `s2_plan`      — tile / block bookkeeping of the dynamic_s2 merge (llava_arch.py:298-390)
  * `dynamic_tile_plan`, `dynamic_preprocess`, `expand2square` — the `dynamic` tiler of the NVILA-Lite recipe and the `pad` mode
    (llava/mm_utils.py:299-338, 505-516)
  * `find_closest_aspect_ratio`, `dynamic_s2_tile_plan`, `dynamic_s2_preprocess` — the dynamic_s2 TILER (llava/mm_utils.py:283-296,
                     341-405): which tile grid an image of a given size gets, and the resize + crop that produces the tiles
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import List, Optional

import torch

from .configs import IGNORE_INDEX


def splice_plan(input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], labels: Optional[torch.Tensor],
                media_lens, media_token_ids, padding_side: str = "right", max_length: Optional[int] = None) -> SimpleNamespace:
    """Row maps for `out[B*S, H]`: text rows come from the embedding table (`txt_src` = token ids -> `txt_dst`), media rows
    from the concatenated media embeddings (row `img_src[i]` -> `img_dst[i]`).  Vectorised: no per-token `.item()`.
    media_lens / media_token_ids: {name: [rows of each embedding block]} / {name: token id} (a plain list + int = {"image": ...});
    the flat media row space is the blocks of every name concatenated in the dict's order, each name consumed like its deque
    (`media_embeds[name].popleft()`, llava_arch.py:462-466) in sample-major order of ITS token.
    max_length: training-time `__truncate_sequence` (llava_arch.py:519-526) — every sample is cut to `model_max_length` AFTER media
    expansion (rows beyond it, text or media, are dropped from the maps)."""
    if not isinstance(media_lens, dict):
        media_lens, media_token_ids = {"image": list(media_lens)}, {"image": int(media_token_ids)}
    ids = input_ids
    B, L = ids.shape
    dev = ids.device
    mask = attention_mask.bool() if attention_mask is not None else torch.ones_like(ids, dtype=torch.bool)
    labels = labels if labels is not None else torch.full_like(ids, IGNORE_INDEX)
    is_img = torch.zeros_like(mask)
    lens = torch.ones_like(ids)
    src0 = torch.zeros_like(ids)
    base, identity = 0, True
    for name, tok in media_token_ids.items():
        blocks = [int(n) for n in media_lens.get(name, [])]
        is_n = (ids == int(tok)) & mask
        n_tok = int(is_n.sum())
        if n_tok < len(blocks):
            raise ValueError(f"Not all {name} embeddings are consumed!")                  # llava_arch.py:481-484
        if n_tok > len(blocks):
            raise IndexError("pop from an empty deque")                                   # media_embeds[name].popleft() on exhausted media
        if blocks:
            ln = torch.tensor(blocks, dtype=torch.long, device=dev)
            pref = torch.cumsum(ln, 0) - ln
            order = (torch.cumsum(is_n.reshape(-1).long(), 0).reshape(B, L) - 1).clamp(min=0)   # j-th block of this name, sample-major
            lens = torch.where(is_n, ln[order], lens)
            src0 = torch.where(is_n, base + pref[order], src0)
            identity = identity and not bool(is_img.any())          # a second name with blocks: flat order != occurrence order in general
            is_img = is_img | is_n
            base += int(ln.sum())
    lens = lens * mask.long()
    n_img = int(is_img.sum())
    ends = torch.cumsum(lens, 1)
    starts = ends - lens
    S_k = ends[:, -1]
    cut = max_length if (max_length is not None and B > 0 and int(S_k.max()) > max_length) else None
    if cut is not None:
        S_k = S_k.clamp(max=cut)
    S = int(S_k.max()) if B > 0 else 0
    right = padding_side == "right"
    row0 = torch.arange(B, device=dev)[:, None] * S + (0 if right else (S - S_k)[:, None])
    dst = row0 + starts
    is_txt = mask & ~is_img
    if cut is not None:
        is_txt = is_txt & (starts < cut)                                              # a text token occupies one position
    txt_src = ids[is_txt].to(torch.int32)
    txt_dst = dst[is_txt].to(torch.int32)
    out_labels = torch.full((B * S,), IGNORE_INDEX, dtype=labels.dtype, device=dev)
    out_labels[txt_dst.long()] = labels[is_txt]
    span = torch.arange(S, device=dev)[None, :]
    out_mask = (span < S_k[:, None]) if right else (span >= (S - S_k)[:, None])
    if n_img:
        lens_occ = lens[is_img]                                                        # occurrence (row-major) order
        total = int(lens_occ.sum())
        offs = torch.arange(total, device=dev) - torch.repeat_interleave(torch.cumsum(lens_occ, 0) - lens_occ, lens_occ)
        img_dst = (torch.repeat_interleave(dst[is_img], lens_occ) + offs).to(torch.int32)
        img_src = (torch.repeat_interleave(src0[is_img], lens_occ) + offs).to(torch.int32)
        if cut is not None:                                                            # media rows beyond the cut are dropped
            keep = (torch.repeat_interleave(starts[is_img], lens_occ) + offs) < cut
            img_dst, img_src = img_dst[keep], img_src[keep]
            identity = False
    else:
        img_dst = torch.zeros(0, dtype=torch.int32, device=dev)
        img_src = torch.zeros(0, dtype=torch.int32, device=dev)
    return SimpleNamespace(B=B, S=S, seqlens=S_k, txt_src=txt_src, txt_dst=txt_dst, img_dst=img_dst, img_src=img_src,
                           img_src_identity=identity, truncated=cut is not None, labels=out_labels.view(B, S), mask=out_mask)


def repack(attention_mask: torch.Tensor, labels: torch.Tensor) -> SimpleNamespace:
    """Packed-row plan: `rows` (flat indices into [B*S] of the kept tokens), restarted `position_ids`, `labels` with the first
    label of every sample masked (:760-762), `cu_seqlens`.  The reference appends one dummy token with mask 0 purely to force
    HF's unpad path (:754-758); `_get_unpad_data` drops it again, so it never reaches the kernels and is not materialised."""
    mask = attention_mask.bool()
    B, S = mask.shape
    seqlens = mask.sum(1).to(torch.int32)
    rows = torch.nonzero(mask.reshape(-1), as_tuple=False).flatten()
    cu = torch.zeros(B + 1, dtype=torch.int32, device=mask.device)
    cu[1:] = torch.cumsum(seqlens, 0)
    seq_id = torch.repeat_interleave(torch.arange(B, device=mask.device), seqlens.long())
    pos = (torch.arange(rows.numel(), device=mask.device) - cu[:-1].long()[seq_id]).to(torch.int32)
    lab = labels.reshape(-1)[rows].clone()
    lab[cu[:-1].long()[seqlens > 0]] = IGNORE_INDEX
    return SimpleNamespace(rows=rows, position_ids=pos, labels=lab, cu_seqlens=cu, seqlens=seqlens,
                           max_seqlen=int(seqlens.max()) if B else 0, seq_of_tok=seq_id.to(torch.int32))


def s2_plan(block_sizes, scales, grid: int, downsample: int, resize_idx: int = -1):
    """Host plan of the dynamic_s2 branch of encode_images (llava/model/llava_arch.py:298-390).  `resize_idx` =
    s2_resize_output_to_scale_idx: the scale whose grid every scale is area-interpolated to (-1 / last = the NVILA recipe,
    scripts/NVILA/stage1_9tile.sh:22; an earlier scale of s x s tiles gives every image s x s output blocks, :346-358).
      desc      [n_blocks, 6] i32  {first tile of the image, bh | obh << 16, bw | obw << 16, block row, block col, single}  for vila_s2_merge_bf16
                (bh x bw = tiles of the image's last scale, obh x obw = its OUTPUT blocks; the high halves are 0 when they are equal)
      n_tiles   total tiles the tower must have produced (checked like the reference's assert, :360-362)
      tile_desc [n_tiles, 8] i32   {first output block of the tile's image, bh | obh << 16, bw | obw << 16, scale index, tile row, tile col,
                single, 0}  for the backward (vila_s2_merge_bwd_bf16)
      perm      per image: for every output token (h w order over the merged (g' obh) x (g' obw) grid, :386-389) its row in the
                projector output [n_blocks * g'^2]   (merge_chessboard + "1 c h w -> (h w) c" as one row gather)
      block_sizes_out  per image (obh, obw) — the reference's `new_block_sizes`
    """
    n_scales = len(scales)
    r = resize_idx % n_scales
    splits = [s // scales[0] for s in scales[:-1]]
    n_pre = sum(s * s for s in splits)
    gd = (grid + downsample - 1) // downsample
    desc, tdesc, perms, out_bs, base, blk = [], [], [], [], 0, 0
    for bs in block_sizes:
        if bs is None:
            desc.append([base, 1, 1, 0, 0, 1])
            tdesc.append([blk, 1, 1, 0, 0, 0, 1, 0])
            perms.append(torch.arange(blk * gd * gd, (blk + 1) * gd * gd, dtype=torch.int32))
            out_bs.append((1, 1))
            base += 1
            blk += 1
            continue
        bh, bw = int(bs[0]), int(bs[1])
        obh, obw = (bh, bw) if r == n_scales - 1 else (splits[r], splits[r])
        w1 = bh | ((obh << 16) if (obh, obw) != (bh, bw) else 0)
        w2 = bw | ((obw << 16) if (obh, obw) != (bh, bw) else 0)
        for i in range(obh):
            for j in range(obw):
                desc.append([base, w1, w2, i, j, 0])
        # tiles of the image in tower order: scales ascending, each scale's chessboard row-major (merge_chessboard, llava_arch.py:255-275)
        for k, sp in enumerate(splits):
            for i in range(sp):
                for j in range(sp):
                    tdesc.append([blk, w1, w2, k, i, j, 0, 0])
        for i in range(bh):
            for j in range(bw):
                tdesc.append([blk, w1, w2, len(splits), i, j, 0, 0])
        Y = torch.arange(gd * obh)[:, None]
        X = torch.arange(gd * obw)[None, :]
        src = (blk + (Y // gd) * obw + (X // gd)) * gd * gd + (Y % gd) * gd + (X % gd)
        perms.append(src.reshape(-1).to(torch.int32))
        out_bs.append((obh, obw))
        base += n_pre + bh * bw
        blk += obh * obw
    return SimpleNamespace(desc=torch.tensor(desc, dtype=torch.int32), tile_desc=torch.tensor(tdesc, dtype=torch.int32), n_tiles=base,
                           n_blocks=blk, perms=perms, splits=splits, block_sizes_out=out_bs)


# ----------------------------------------------------------------------------------------------------------------------
# dynamic_s2 tiler (host half): llava/mm_utils.py:283-296 (find_closest_aspect_ratio), 341-405 (dynamic_s2_preprocess)
# ----------------------------------------------------------------------------------------------------------------------
def find_closest_aspect_ratio(aspect_ratio: float, target_ratios, width: int, height: int, image_size: int):
    """mm_utils.py:283-296: the (cols, rows) grid whose cols / rows is closest to the image's width / height; on an exact tie the LATER
    candidate wins only if the image has more than half the pixels of that grid (so small images keep the smaller grid)."""
    best_diff, best = float("inf"), (1, 1)
    area = width * height
    for ratio in target_ratios:
        diff = abs(aspect_ratio - ratio[0] / ratio[1])
        if diff < best_diff:
            best_diff, best = diff, ratio
        elif diff == best_diff and area > 0.5 * image_size * image_size * ratio[0] * ratio[1]:
            best = ratio
    return best


def dynamic_s2_tile_plan(width: int, height: int, s2_scales=(448, 896, 1344), max_num: int = 12, image_size: int = 448):
    """The integer half of `dynamic_s2_preprocess` (mm_utils.py:341-405) -> (resizes, block_size).
    resizes = [((target_w, target_h), [crop boxes (l, t, r, b) in tile order])] — one entry per scale: every scale but the last resizes the
    image to a SQUARE of (scale / scales[0])^2 tiles; the last scale picks the (cols, rows) grid with min_num <= cols * rows <= max_num,
    min_num = (scales[-1] // scales[0])^2, closest to the image's aspect ratio.  block_size = (rows, cols) of that last grid — what
    `merge_features_for_dynamic_s2` receives (llava_arch.py:298-364)."""
    scales = list(s2_scales)
    aspect = width / height
    min_num = (scales[-1] // scales[0]) ** 2

    def boxes(tw: int, th: int):
        per_row = tw // image_size
        n = per_row * (th // image_size)
        return [((i % per_row) * image_size, (i // per_row) * image_size, (i % per_row + 1) * image_size, (i // per_row + 1) * image_size)
                for i in range(n)]
    resizes = []
    for scale in scales[:-1]:
        k = scale // scales[0]
        resizes.append(((image_size * k, image_size * k), boxes(image_size * k, image_size * k)))
    ratios = {(i, j) for n in range(min_num, max_num + 1) for i in range(1, n + 1) for j in range(1, n + 1) if min_num <= i * j <= max_num}
    ratios = sorted(ratios, key=lambda x: x[0] * x[1])            # stable: ties keep the set's iteration order, as in the reference
    cols, rows = find_closest_aspect_ratio(aspect, ratios, width, height, image_size)
    resizes.append(((image_size * cols, image_size * rows), boxes(image_size * cols, image_size * rows)))
    return resizes, (rows, cols)


def dynamic_s2_preprocess(image, s2_scales=(448, 896, 1344), max_num: int = 12, image_size: int = 448):
    """mm_utils.py:341-405 on a PIL image -> (tiles: list of PIL images of image_size^2, block_size (rows, cols)).  `Image.resize` with its
    default filter (bicubic) exactly as the reference calls it, then the crops of the plan."""
    w, h = image.size
    resizes, block_size = dynamic_s2_tile_plan(w, h, s2_scales, max_num, image_size)
    tiles = []
    for size, boxes in resizes:
        resized = image.resize(size)
        tiles.extend(resized.crop(b) for b in boxes)
    return tiles, block_size


# ----------------------------------------------------------------------------------------------------------------------
# `dynamic` tiler (every NVILA-Lite script: `--image_aspect_ratio dynamic`): llava/mm_utils.py:299-338 (dynamic_preprocess), :505-518 (pad)
# ----------------------------------------------------------------------------------------------------------------------
def dynamic_tile_plan(width: int, height: int, min_num: int = 1, max_num: int = 12, image_size: int = 384, use_thumbnail: bool = True):
    """The integer half of `dynamic_preprocess` (mm_utils.py:299-338) -> ((target_w, target_h), crop boxes in tile order, thumbnail?).
    The (cols, rows) grid with min_num <= cols * rows <= max_num closest to the image's aspect ratio (`find_closest_aspect_ratio`), tiles
    walked row-major; a whole-image thumbnail follows the tiles unless the grid is 1 x 1."""
    ratios = {(i, j) for n in range(min_num, max_num + 1) for i in range(1, n + 1) for j in range(1, n + 1) if min_num <= i * j <= max_num}
    ratios = sorted(ratios, key=lambda x: x[0] * x[1])
    cols, rows = find_closest_aspect_ratio(width / height, ratios, width, height, image_size)
    tw, th = image_size * cols, image_size * rows
    boxes = [((i % cols) * image_size, (i // cols) * image_size, (i % cols + 1) * image_size, (i // cols + 1) * image_size)
             for i in range(cols * rows)]
    return (tw, th), boxes, bool(use_thumbnail and len(boxes) != 1)


def dynamic_preprocess(image, min_num: int = 1, max_num: int = 12, image_size: int = 384, use_thumbnail: bool = True):
    """mm_utils.py:299-338 on a PIL image -> list of PIL tiles of image_size^2 (+ the thumbnail).  `Image.resize` with its default filter,
    exactly as the reference calls it."""
    w, h = image.size
    size, boxes, thumb = dynamic_tile_plan(w, h, min_num, max_num, image_size, use_thumbnail)
    resized = image.resize(size)
    tiles = [resized.crop(b) for b in boxes]
    if thumb:
        tiles.append(image.resize((image_size, image_size)))
    return tiles


def expand2square(pil_img, background_color):
    """mm_utils.py:505-516 (`image_aspect_ratio == "pad"`): the picture centred on a square canvas of its longer side."""
    from PIL import Image
    width, height = pil_img.size
    if width == height:
        return pil_img
    side = max(width, height)
    result = Image.new(pil_img.mode, (side, side), background_color)
    result.paste(pil_img, (0, (width - height) // 2) if width > height else ((height - width) // 2, 0))
    return result
