"""FlatCluster restatement used by examples/hinit_model.py (HTrain.c:763-803): structural properties on CPU; its numbers are pinned by the
HInit mixture fixtures on the GPU box (tests/test_gpu_demo.py::test_hinit_mixture_training)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def test_flat_cluster_partitions_the_pool():
    from examples.hinit_model import flat_cluster, MIN_CLUST_SIZE
    rng = np.random.default_rng(3)
    X = np.concatenate([rng.normal(c, 0.3, size=(40, 5)) for c in (-3.0, 0.0, 4.0)]).astype(np.float32)
    rng.shuffle(X)
    size, ctr, var = flat_cluster(X, 3)
    assert size.sum() == X.shape[0] and (size >= MIN_CLUST_SIZE).all()
    assert sorted(np.round(ctr.mean(axis=1)).tolist()) == [-3.0, 0.0, 4.0]          # the three blobs are found
    assert (var > 0).all() and (var < 0.5).all()
    # one cluster: the pool's mean and (biased) variance
    s1, c1, v1 = flat_cluster(X, 1)
    assert s1.tolist() == [X.shape[0]]
    assert np.allclose(c1[0], X.mean(0), rtol=1e-5, atol=1e-6) and np.allclose(v1[0], X.var(0), rtol=1e-4)


def test_flat_cluster_refuses_too_few_items():
    import pytest
    from examples.hinit_model import flat_cluster
    with pytest.raises(ValueError):
        flat_cluster(np.zeros((2, 4), np.float32), 3)
