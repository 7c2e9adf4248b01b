"""Egocentric map frames for `VIDEO_OPTION` runs (`occupancy_map_viz`, `semantic_map_viz`; behaviour of
ivlnce_baselines/common/mapping_module/visualize_semantic_map.py:13-125).  Host-side numpy, off the hot path:
only produced when a video option is set.  The class palette is the reference's; the upscale is the same
nearest-neighbour resize to 200 px; the agent marker is a drawn arrow head - habitat's AGENT_SPRITE image (and
cv2 / imutils) are not available here, so frames are not pixel-identical around the marker."""
import numpy as np
import torch

LABEL_COLOURS = np.array([
    (0, 0, 0), (106, 137, 204), (230, 126, 34), (7, 153, 146), (248, 194, 145), (76, 209, 55), (255, 168, 1),
    (184, 233, 148), (39, 174, 96), (229, 80, 57), (30, 55, 153), (24, 220, 255), (234, 32, 39),
], dtype=np.uint8)  # void, shelving, chest of drawers, bed, cushion, fireplace, sofa, table, chair, cabinet, plant, counter, sink


def _resize_nearest(img: np.ndarray, width: int) -> np.ndarray:
    h, w = img.shape[:2]
    height = int(round(h * width / float(w)))
    rows = np.minimum((np.arange(height) * (h / height)).astype(np.int64), h - 1)
    cols = np.minimum((np.arange(width) * (w / width)).astype(np.int64), w - 1)
    return img[rows][:, cols]


def _draw_agent(img: np.ndarray) -> np.ndarray:
    """Marker at the map centre (the agent's position in an egocentric map), pointing up = forward."""
    h, w = img.shape[:2]
    size = max(3, int(0.05 * max(h, w)))
    r0, c0 = h // 2 - size // 2, w // 2
    for i in range(size):
        half = (i * size) // (2 * size) + (i // 2)
        img[r0 + i, max(0, c0 - half): c0 + half + 1] = (255, 255, 255) if i < size - 1 else (128, 128, 128)
    return img


def _frames(maps, colour_fn, min_width=200):
    out = []
    for m in (maps.cpu().numpy() if torch.is_tensor(maps) else np.asarray(maps)):
        out.append(_draw_agent(_resize_nearest(colour_fn(m), min_width).copy()))
    return np.stack(out)


def visualize_ego_occupancy_map(occupancy_map):
    """(B,R,C) {0,1} -> (B,200,200,3) u8: free = white, occupied = black (visualize_semantic_map.py:56-60)."""
    return _frames(occupancy_map, lambda m: np.repeat((255 - m.astype(np.int32) * 255).astype(np.uint8)[:, :, None], 3, 2))


def visualize_ego_semantic_map(semantic_map):
    """(B,R,C) labels 0..12 -> (B,200,200,3) u8 through the class palette (:30-53)."""
    return _frames(semantic_map, lambda m: LABEL_COLOURS[m.astype(np.int64)])
