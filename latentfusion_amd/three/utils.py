"""Farthest-point sampling (latentfusion/three/utils.py:4-49)."""
import torch


def farthest_points(data, n_clusters: int, dist_func, return_center_indexes=False, return_distances=False, verbose=False):
    """Greedy farthest-point clustering: returns the cluster index of every row [, centre indices][, distances].
    The first centre is row 0 (argmax over the constant initial distances), like the reference."""
    n = data.shape[0]
    if n_clusters >= n:
        ar = torch.arange(n, dtype=torch.long)
        return (ar, ar.clone()) if return_center_indexes else ar
    clusters = torch.full((n,), -1, dtype=torch.long)
    distances = torch.full((n,), 1e7, dtype=torch.float32)
    centers = torch.zeros(n_clusters, dtype=torch.long)
    for i in range(n_clusters):
        c = torch.argmax(distances)
        centers[i] = c
        new = dist_func(data[c].unsqueeze(0).expand(n, -1), data)
        distances = torch.min(distances, new)
        clusters[distances == new] = i
        if verbose:
            print('farthest points max distance : {}'.format(torch.max(distances)))
    if return_center_indexes:
        return (clusters, centers, distances) if return_distances else (clusters, centers)
    return clusters
