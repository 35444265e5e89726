"""Greedy farthest-point sampling / clustering (latentfusion/three/utils.py:4-49)."""
import torch


def farthest_points(data, n_clusters: int, dist_func, return_center_indexes=False, return_distances=False, verbose=False):
    """Picks `n_clusters` rows of `data`, each the farthest from those already picked (the first is row 0, the
    arg-max of the constant initial distances), and assigns every row to the LAST centre that attains its
    minimum distance.  Returns cluster ids [, centre indices [, distances to the nearest centre]]."""
    count = data.shape[0]
    if n_clusters >= count:                                  # every row is its own centre
        ids = torch.arange(count, dtype=torch.long)
        return (ids, ids.clone()) if return_center_indexes else ids
    owner = torch.full((count,), -1, dtype=torch.long)
    nearest = torch.full((count,), 1e7, dtype=torch.float32)
    picked = []
    for k in range(n_clusters):
        picked.append(int(torch.argmax(nearest)))
        to_new = dist_func(data[picked[-1]].unsqueeze(0).expand(count, -1), data)
        nearest = torch.minimum(nearest, to_new)
        owner[nearest == to_new] = k
        if verbose:
            print('farthest points max distance : {}'.format(nearest.max()))
    centers = torch.tensor(picked, dtype=torch.long)
    if not return_center_indexes:
        return owner
    return (owner, centers, nearest) if return_distances else (owner, centers)
