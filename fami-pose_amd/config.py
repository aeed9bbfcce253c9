"""Minimal config node: attribute AND item access, as the reference reads its
yacs tree both ways (`cfg.MODEL.EXTRA` hrnet.py:571 and
`cfg['MODEL']['EXTRA']['STAGE2']` hrnet.py:590).  yacs itself is not a
dependency.  `default_cfg` carries the keys the model reads
with the values of configs/Alignment/Base_PoseTrack17.yaml:28-87 and
configs/Alignment/posetrack17/Alignment_V15.yaml."""


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(d):
        if isinstance(d, dict) and not isinstance(d, CfgNode):
            return CfgNode({k: CfgNode.wrap(v) for k, v in d.items()})
        return d


def default_cfg(width=48, num_joints=17, freeze_backbone=False, name='Alignment_V15', image_size=(288, 384),
                num_sup=4, dcn_groups=None):
    w = int(width)
    hm = (image_size[0] // 4, image_size[1] // 4)
    return CfgNode.wrap({
        'MODEL': {
            'NAME': name, 'NUM_JOINTS': num_joints, 'INIT_WEIGHTS': True, 'PRETRAINED': '', 'BACKBONE_PRETRAINED': '',
            'FREEZE_HRNET_WEIGHTS': bool(freeze_backbone), 'IMAGE_SIZE': list(image_size), 'HEATMAP_SIZE': list(hm),
            'SIGMA': 3, 'NUM_SUPPORT_FRAMES': num_sup, 'DCN_OFFSET_GROUPS': dcn_groups,
            'EXTRA': {
                'FINAL_CONV_KERNEL': 1,
                'STAGE2': {'NUM_MODULES': 1, 'NUM_BRANCHES': 2, 'BLOCK': 'BASIC', 'NUM_BLOCKS': [4, 4],
                           'NUM_CHANNELS': [w, 2 * w], 'FUSE_METHOD': 'SUM'},
                'STAGE3': {'NUM_MODULES': 4, 'NUM_BRANCHES': 3, 'BLOCK': 'BASIC', 'NUM_BLOCKS': [4, 4, 4],
                           'NUM_CHANNELS': [w, 2 * w, 4 * w], 'FUSE_METHOD': 'SUM'},
                'STAGE4': {'NUM_MODULES': 3, 'NUM_BRANCHES': 4, 'BLOCK': 'BASIC', 'NUM_BLOCKS': [4, 4, 4, 4],
                           'NUM_CHANNELS': [w, 2 * w, 4 * w, 8 * w], 'FUSE_METHOD': 'SUM'}}},
        'DATASET': {'BBOX_ENLARGE_FACTOR': 1.25},
        'TRAIN': {'ROT_FACTOR': 45, 'SCALE_FACTOR': 0.35, 'LR': 1e-3, 'LR_STEP': [8, 12, 16], 'LR_FACTOR': 0.1,
                  'OPTIMIZER': 'adam'},
        'LOSS': {'HEATMAP_MSE': {'USE': True, 'WEIGHT': 1.0}},
    })
