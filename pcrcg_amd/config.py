"""Configuration of the hot path.  Keys and defaults mirror the reference's flattened YAML
(ref:lib/utils.py:46-65, ref:configs/test/indoor.yaml:16-50, ref:configs/test/kitti.yaml:10-34) and
its block lists (ref:configs/models.py:1-57)."""


class Config(dict):
    """Flat dict with attribute access (the reference wraps its config in easydict.EasyDict,
    ref:main.py:22-23)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    def copy(self):
        return Config(self)


_ENC_DEC = ["simple", "resnetb", "resnetb_strided", "resnetb", "resnetb", "resnetb_strided", "resnetb", "resnetb",
            "resnetb_strided", "resnetb", "resnetb", "nearest_upsample", "unary", "nearest_upsample", "unary",
            "nearest_upsample", "last_unary"]

architectures = {
    "indoor": list(_ENC_DEC),
    "kitti": list(_ENC_DEC),
    "modelnet": ["simple", "resnetb", "resnetb", "resnetb_strided", "resnetb", "resnetb", "resnetb_strided",
                 "resnetb", "resnetb", "nearest_upsample", "unary", "unary", "nearest_upsample", "unary",
                 "last_unary"],
}

_COMMON = dict(
    num_layers=4, in_points_dim=3, first_feats_dim=256, final_feats_dim=32, in_feats_dim=1,
    deform_radius=5.0, num_kernel_points=15, KP_extent=2.0, KP_influence="linear", aggregation_mode="sum",
    fixed_kernel_points="center", use_batch_norm=True, batch_norm_momentum=0.02, deformable=False,
    modulated=False, dgcnn_k=10, num_head=4, nets=["self", "cross", "self"],
    # PCR-CG's 2-D branch and auxiliary heads are outside this path
    image_feature=False, img_num=0, init_mode="", node_overlap=False, quaternion=False,
)


def indoor_config(**over):
    """3DMatch / 3DLoMatch geometry-only configuration (BASELINE.json configs[0..3])."""
    cfg = Config(_COMMON, dataset="indoor", first_subsampling_dl=0.025, conv_radius=2.5, gnn_feats_dim=512,
                 overlap_radius=0.0375)
    cfg.update(over)
    cfg["architecture"] = list(architectures[cfg["dataset"]])
    return cfg


def kitti_config(**over):
    """KITTI odometry configuration (BASELINE.json configs[4])."""
    cfg = Config(_COMMON, dataset="kitti", first_subsampling_dl=0.3, conv_radius=4.25, gnn_feats_dim=256,
                 overlap_radius=0.45)
    cfg.update(over)
    cfg["architecture"] = list(architectures[cfg["dataset"]])
    return cfg


def modelnet_config(**over):
    """ModelNet40 configuration (ref:configs/test/modelnet.yaml:12-35): three levels, the `modelnet` block list with two
    consecutive unary blocks in the decoder (ref:configs/models.py:42-57)."""
    cfg = Config(_COMMON, dataset="modelnet", num_layers=3, first_feats_dim=512, final_feats_dim=96, first_subsampling_dl=0.06,
                 conv_radius=2.75, gnn_feats_dim=256, overlap_radius=0.04)
    cfg.update(over)
    cfg["architecture"] = list(architectures[cfg["dataset"]])
    return cfg


def as_config(cfg):
    return cfg if isinstance(cfg, Config) else Config(cfg)
