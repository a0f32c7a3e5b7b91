"""Data wire formats on either side of the detector.

In:  `collate_kitti` -- the batch layout the reference's DataLoader hands to `example_to_device`
     (det3d/torchie/parallel/collate.py:91-171).  Pinned: tests/golden/collate.json holds the output
     of the reference's own function on a seeded batch.
Out: detections -> nuScenes result records (det3d/datasets/nuscenes/nusc_common.py:160-201
     `_second_det_to_nusc_box`, `_lidar_nusc_box_to_global`; det3d/datasets/nuscenes/nuscenes.py:
     227-277 the per-box JSON record and attribute rule), so a results file can be written for the
     nuScenes devkit when data exists.  The reference builds these with `pyquaternion.Quaternion` and
     nuscenes-devkit's `Box`, neither of which is installed here: the quaternion algebra below
     follows their published definitions (Hamilton product, (w, x, y, z), active rotations) and is
     checked against rotation matrices -- parity unpinned.
"""
import collections
import operator

import numpy as np
import torch

_CAT0 = ("voxels", "num_points", "num_gt", "voxel_labels", "num_voxels", "cyv_voxels", "cyv_num_points",
         "cyv_num_voxels")
_PER_TASK = ("anchors", "anchors_mask", "reg_targets", "reg_weights", "labels", "hm", "anno_box", "ind", "mask",
             "cat", "gt_box")


def collate_kitti(batch_list, samples_per_gpu=1):
    """collate.py:91-171.  `points` stays a list of per-sample tensors (the detectors take it as
    `example["points"]`); per-task targets are stacked over the batch, one tensor per task."""
    merged = collections.defaultdict(list)
    for example in batch_list:
        for sub in (example if type(example) is list else [example]):
            for k, v in sub.items():
                merged[k].append(v)
    batch_size = len(merged["metadata"])
    ret = {}
    for key, elems in merged.items():
        if key in _CAT0:
            ret[key] = torch.tensor(np.concatenate(elems, axis=0))
        elif key == "gt_boxes":
            res = []
            for task_id in range(len(elems[0])):
                max_gt = max(len(elems[k][task_id]) for k in range(batch_size))
                out = np.zeros((batch_size, max_gt, 7))
                for i in range(batch_size):
                    out[i, : len(elems[i][task_id]), :] = elems[i][task_id]
                res.append(out)
            ret[key] = res
        elif key == "metadata":
            ret[key] = elems
        elif key == "calib":
            calib = {}
            for elem in elems:
                for k1, v1 in elem.items():
                    calib.setdefault(k1, []).append(v1)
            ret[key] = {k1: torch.tensor(np.stack(v1, axis=0)) for k1, v1 in calib.items()}
        elif key == "points":
            ret[key] = [torch.tensor(elem) for elem in elems]
        elif key in ("coordinates", "cyv_coordinates"):
            ret[key] = torch.tensor(np.concatenate(
                [np.pad(c, ((0, 0), (1, 0)), mode="constant", constant_values=i) for i, c in enumerate(elems)], axis=0))
        elif key in _PER_TASK:
            per_task = collections.defaultdict(list)
            for elem in elems:
                for idx, ele in enumerate(elem):
                    per_task[idx].append(torch.tensor(ele))
            ret[key] = [torch.stack(v) for v in per_task.values()]
        elif key == "raw_gt_box":
            per_task = collections.defaultdict(list)
            for elem in elems:
                for idx, ele in enumerate(elem):
                    per_task[idx].append(torch.tensor(ele))
            ret[key] = list(per_task.values())
        elif key == "gt_boxes_and_cls":
            ret[key] = torch.tensor(np.stack(elems, axis=0))
        else:
            ret[key] = np.stack(elems, axis=0)
    return ret


# ---- detections -> nuScenes records ------------------------------------------------------------------
def quat_about_z(radians):
    """Quaternion(axis=[0, 0, 1], radians=r) as (w, x, y, z)."""
    return np.array([np.cos(radians / 2.0), 0.0, 0.0, np.sin(radians / 2.0)], dtype=np.float64)


def quat_mul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dtype=np.float64)


def quat_matrix(q):
    w, x, y, z = np.asarray(q, np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class NuscBox:
    """The fields of nuscenes.utils.data_classes.Box that the export uses, with its rotate / translate."""

    def __init__(self, center, size, orientation, label, score, velocity):
        self.center = np.asarray(center, np.float64).copy()
        self.wlh = np.asarray(size, np.float64).copy()
        self.orientation = np.asarray(orientation, np.float64).copy()
        self.label, self.score = int(label), float(score)
        self.velocity = np.asarray(velocity, np.float64).copy()

    def rotate(self, q):
        R = quat_matrix(q)
        self.center = R @ self.center
        self.orientation = quat_mul(np.asarray(q, np.float64), self.orientation)
        self.velocity = R @ self.velocity

    def translate(self, t):
        self.center = self.center + np.asarray(t, np.float64)


def second_det_to_nusc_box(detection):
    """nusc_common.py:160-178: heading -> -heading - pi/2 (det3d's lidar frame to the nuScenes box frame),
    one box per detection; the box's size is handed over as stored, (dx, dy, dz)."""
    box3d = detection["box3d_lidar"].detach().cpu().numpy().copy()
    scores = detection["scores"].detach().cpu().numpy()
    labels = detection["label_preds"].detach().cpu().numpy()
    box3d[:, -1] = -box3d[:, -1] - np.pi / 2
    return [NuscBox(box3d[i, :3], box3d[i, 3:6], quat_about_z(box3d[i, -1]), labels[i], scores[i],
                    (*box3d[i, 6:8], 0.0)) for i in range(box3d.shape[0])]


def lidar_nusc_box_to_global(boxes, cs_record, pose_record):
    """nusc_common.py:181-201 with the two database records passed in: lidar -> ego (calibrated sensor
    rotation / translation) -> global (ego pose)."""
    for box in boxes:
        box.rotate(cs_record["rotation"])
        box.translate(cs_record["translation"])
        box.rotate(pose_record["rotation"])
        box.translate(pose_record["translation"])
    return boxes


_MOVING = ("car", "construction_vehicle", "bus", "truck", "trailer")


def nusc_annos(detections, class_names, records, cls_attr_dist, name_mapping=None):
    """nuscenes.py:213-277: `{"results": {token: [record, ...]}, "meta": {...}}`.
    detections: iterable of detector outputs whose metadata carries `token`; records: token ->
    (calibrated_sensor record, ego_pose record); cls_attr_dist: class -> {attribute: count} (the
    table at nusc_common.py:47-158, dataset statistics that travel with the dataset tooling)."""
    names = [(name_mapping or {}).get(n, n) for n in class_names]
    results = {}
    for det in detections:
        token = det["metadata"]["token"]
        boxes = lidar_nusc_box_to_global(second_det_to_nusc_box(det), *records[token])
        annos = []
        for box in boxes:
            name = names[box.label]
            if np.sqrt(box.velocity[0] ** 2 + box.velocity[1] ** 2) > 0.2:
                attr = "vehicle.moving" if name in _MOVING else ("cycle.with_rider" if name in ("bicycle", "motorcycle") else None)
            else:
                attr = "pedestrian.standing" if name == "pedestrian" else ("vehicle.stopped" if name == "bus" else None)
            if attr is None:
                attr = max(cls_attr_dist[name].items(), key=operator.itemgetter(1))[0]
            annos.append({"sample_token": token, "translation": box.center.tolist(), "size": box.wlh.tolist(),
                          "rotation": box.orientation.tolist(), "velocity": box.velocity[:2].tolist(),
                          "detection_name": name, "detection_score": box.score, "attribute_name": attr})
        results[token] = annos
    return {"results": results, "meta": {"use_camera": False, "use_lidar": True, "use_radar": False,
                                         "use_map": False, "use_external": False}}
