"""pcrcg_amd -- MI355X-native implementation of PCR-CG's feature-extraction hot path:
grid subsampling + radius neighbours (front end), the KPConv encoder/decoder and the GNN overlap
head, behind the reference's own interfaces.  See DESIGN.md and INTEGRATION.md."""
from .config import Config, indoor_config, kitti_config, modelnet_config, architectures  # noqa: F401

__all__ = ["Config", "indoor_config", "kitti_config", "modelnet_config", "architectures"]
