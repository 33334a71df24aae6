"""gnn_manip_amd: MI355X-native (gfx950) rollout engine for gnn-manip's encode-process-decode
particle simulator.  HIP kernels + C ABI in libgnnmanip_hip.so; this package is the host-side
mirror of the reference's call surface (see DESIGN.md / INTEGRATION.md)."""
from .epd_gnn import EncProcDecGNN, GraphIndependent, InteractionNetwork  # noqa: F401
from .dataset import CoffeeDataset, CoffeeTestDataset, GraphData, GraphLoader, collate_graphs, read_metadata  # noqa: F401
from .graph import (GraphBoundedMultimaterial, GraphBoundedMultimaterialControl, compute_acceleration,  # noqa: F401
                    get_connectivity, get_edges_displacement, random_walk_noise)
from .rollout import RolloutEngine, get_position_from_prediction  # noqa: F401
