"""`set_by_task_cfg` (det3d/core/utils/center_utils.py:229-274): per-class test_cfg lists are
regrouped into per-task lists; scalars are shared by every task."""


def set_by_task_cfg(test_cfg, task_num_classes):
    total = sum(task_num_classes)

    def regroup(param):
        if isinstance(param, (float, int)):
            return param
        assert isinstance(param, (list, tuple))
        assert len(param) == total
        out, start = [], 0
        for n in task_num_classes:
            out.append(list(param[start:start + n]))
            start += n
        return out

    if test_cfg.get("rectifier", False):
        test_cfg["rectifier"] = regroup(test_cfg["rectifier"])
    if test_cfg.get("use_rectify", False):
        test_cfg["use_rectify"] = regroup(test_cfg["use_rectify"])
    nms = test_cfg["nms"]
    for key in ("nms_pre_max_size", "nms_post_max_size", "nms_iou_threshold"):
        nms[key] = regroup(nms[key])
    return test_cfg
