"""GPU probe: why does (or does not) the cluster-bound pruning engage?  prints the index geometry and the prediction."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.conftest import gmm
from torchdr_amd.distance import base as dbase

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
X = gmm(n, d, 2.0).cuda()
P = dbase.PackedPoints(X)
ops = dbase._screen_operands(P, P)
tier, tau = dbase._choose_tier(P, P, ops, 0, 30, "sqeuclidean", True, 0)
ci = dbase._cluster_index(P, ops)
gap = (ci.dist - ci.radius[:, None] - ci.radius[None, :]).clamp(min=0)
off = ~torch.eye(ci.n_clusters, dtype=torch.bool, device=gap.device)
print(json.dumps({
    "n": n, "d": d, "tier": tier, "pilot_tau": tau, "clusters": ci.n_clusters,
    "radius_mean": float(ci.radius.mean()), "radius_max": float(ci.radius.max()),
    "centre_dist_min": float(ci.dist[off].min()), "centre_dist_median": float(ci.dist[off].median()),
    "gap_zero_share": float((gap[off] == 0).float().mean()),
    "scan_fraction(tau)": ci.scan_fraction(tau), "scan_fraction(2tau)": ci.scan_fraction(2 * tau),
    "tiles_min": int(ci.tiles.min()), "tiles_max": int(ci.tiles.max()),
}))
