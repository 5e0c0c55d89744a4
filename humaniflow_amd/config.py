"""Configuration with the reference's attribute names.

The reference uses a yacs ``CfgNode`` (configs/humaniflow_config.py:5-108); yacs is not a
dependency here, so ``CfgNode`` below is a small attribute-dict with the same access
pattern (``cfg.MODEL.NORM_FLOW.CONTEXT_DIM``), ``clone()``, ``merge_from_list()`` (the
reference's ``-O KEY VAL`` override, scripts/run_train.py:43-44) and YAML dump/load.
Only the MODEL and DATA groups matter for the inference hot path; TRAIN/LOSS values that
the path's callers read are kept for drop-in compatibility.
"""
import copy
import math

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def merge_from_list(self, kv):
        assert len(kv) % 2 == 0
        for key, val in zip(kv[0::2], kv[1::2]):
            node = self
            parts = key.split('.')
            for p in parts[:-1]:
                node = node[p]
            old = node[parts[-1]]
            if isinstance(val, str) and not isinstance(old, str):
                val = yaml.safe_load(val)
            node[parts[-1]] = type(old)(val) if old is not None and not isinstance(old, (list, type(None))) else val

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f))

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and k in self:
                self[k]._merge(v)
            else:
                self[k] = v

    def to_plain(self):
        return {k: (v.to_plain() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self):
        return yaml.safe_dump(self.to_plain())


def _node(**kw):
    n = CfgNode()
    for k, v in kw.items():
        n[k] = v
    return n


def get_humaniflow_cfg_defaults():
    """Same defaults as configs/humaniflow_config.py:5-108 (MODEL/DATA groups, plus the
    TRAIN/LOSS scalars the inference callers touch)."""
    C = CfgNode()
    C.MODEL = _node(NUM_IN_CHANNELS=18, NUM_RESNET_LAYERS=18, INPUT_SHAPE_GLOB_CAM_FEATS_DIM=256, NUM_SMPL_BETAS=10)
    C.MODEL.NORM_FLOW = _node(CONTEXT_DIM=64, NUM_TRANSFORMS=2, TRANSFORM_TYPE='spline_coupling',
                              TRANSFORM_NN_HIDDEN_DIMS=[64, 32, 32], NUM_SPLINE_SEGMENTS=8,
                              PERMUTE_TYPE='permute', PERMUTE_NN_HIDDEN_DIMS=None,
                              COMPACT_SUPPORT_RADIUS=1.5 * math.pi, BASE_DIST_STD=0.6)
    C.DATA = _node(BBOX_THRESHOLD=0.95, BBOX_SCALE_FACTOR=1.2, PROXY_REP_SIZE=256, HEATMAP_GAUSSIAN_STD=4.0,
                   EDGE_NMS=True, EDGE_THRESHOLD=0.0, EDGE_GAUSSIAN_STD=1.0, EDGE_GAUSSIAN_SIZE=5)
    C.TRAIN = _node(NUM_EPOCHS=231, BATCH_SIZE=72, LR=0.0001, EPOCHS_PER_SAVE=5, PIN_MEMORY=True, NUM_WORKERS=2)
    C.TRAIN.SYNTH_DATA = _node(FOCAL_LENGTH=300.0, MEAN_CAM_T=[0.0, -0.2, 2.5])
    C.TRAIN.SYNTH_DATA.AUGMENT = CfgNode()
    C.TRAIN.SYNTH_DATA.AUGMENT.SMPL = _node(SHAPE_STD=1.25)
    C.LOSS = _node(REDUCTION='mean', J2D_LOSS_ON='point_est+samples', NUM_J2D_SAMPLES=8, APPLY_POINT_EST_LOSS=False)
    C.LOSS.WEIGHTS = _node(POSE=150.0, SHAPE=50.0, JOINTS2D=25000.0, GLOB_ROTMATS=5000.0, VERTS3D=0.0, JOINTS3D=0.0)
    return C


def get_benchmark_cfg():
    """BASELINE.json configs[1]: ResNet-50 encoder, everything else default."""
    C = get_humaniflow_cfg_defaults()
    C.MODEL.NUM_RESNET_LAYERS = 50
    return C


# configs/paths.py:1-5 (model files).  The three extra-joint regressors ship with this package
# in sparse form (see humaniflow_amd/model_files/README.md); SMPL itself is licensed and absent.
SMPL = './model_files/smpl'
