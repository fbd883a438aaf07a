"""An INDEPENDENT statement of linear blend skinning  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

`smplx.lbs.lbs` is a third-party dependency of the reference (pinned `smplx==0.1.26`, called at
model_training/model/flame.py:212-221) whose source is not under /root/reference; `oracle/flame_ref.py` restates it from
the published implementation. This module is the cross-check the restatement is held to: it is written from the SMPL
PAPER (Loper et al., "SMPL: A Skinned Multi-Person Linear Model", SIGGRAPH Asia 2015, equations 2-10), in float64, one
vertex and one joint at a time, with scipy's rotation-vector exponential instead of a hand-written Rodrigues -- it shares no
code, no vectorisation and no operation order with `flame_ref.lbs`:

    eq. 8/9   shape blend shapes        T_s = T_bar + sum_n beta_n S_n
    eq. 10    joints                    J   = Jreg . T_s
    eq. 9     pose blend shapes         T_p = T_s + sum_n (R_n(theta) - R_n(theta*)) P_n      (theta* = rest pose: identity;
                                        n runs over the 9 K elements of the K non-root joint rotations)
    eq. 3/4   world transforms          G_k = prod_{j in A(k)} [ exp(w_j) | j_j ; 0 1 ]        (j_j relative to the parent)
              rest-pose removal         G'_k = G_k . [ I | -J_k ; 0 1 ]
    eq. 2     skinning                  t'_i = sum_k w_{k,i} G'_k [t_i ; 1]

Differences from the oracle it is compared with are rounding only (float32 there) plus the 1e-8 the published code adds to
the rotation vector before taking its norm.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation


def lbs_per_vertex(betas, pose, v_template, shapedirs, posedirs, j_regressor, parents, weights):
    """One image. betas [L], pose [K+1,3] axis-angle (row 0 = root), v_template [V,3], shapedirs [V,3,L], posedirs
    [9K, 3V] (row n = P_n flattened vertex-major), j_regressor [K+1,V], parents [K+1] (root: -1), weights [V,K+1].
    Returns (vertices [V,3], posed joint locations [K+1,3]) in float64."""
    betas = np.asarray(betas, np.float64)
    pose = np.asarray(pose, np.float64)
    v_template, shapedirs = np.asarray(v_template, np.float64), np.asarray(shapedirs, np.float64)
    posedirs, j_regressor = np.asarray(posedirs, np.float64), np.asarray(j_regressor, np.float64)
    weights = np.asarray(weights, np.float64)
    nv, nj = v_template.shape[0], j_regressor.shape[0]
    rot = [Rotation.from_rotvec(pose[k]).as_matrix() for k in range(nj)]   # exp(w_k)
    # pose blend-shape coefficients: (R_n(theta) - I), joints 1..K, row-major 3x3 each
    coeff = np.concatenate([(rot[k] - np.eye(3)).reshape(9) for k in range(1, nj)])
    # shape blend shapes, vertex by vertex
    t_s = np.empty((nv, 3))
    for i in range(nv):
        t_s[i] = v_template[i] + shapedirs[i] @ betas
    joints = np.empty((nj, 3))
    for k in range(nj):
        joints[k] = sum(j_regressor[k, i] * t_s[i] for i in np.nonzero(j_regressor[k])[0])
    # world transforms down the kinematic tree
    G = [None] * nj
    for k in range(nj):
        local = np.eye(4)
        local[:3, :3] = rot[k]
        local[:3, 3] = joints[k] - (joints[parents[k]] if parents[k] >= 0 else 0.0)
        G[k] = local if parents[k] < 0 else G[parents[k]] @ local
    posed_joints = np.stack([G[k][:3, 3] for k in range(nj)])
    Gp = []
    for k in range(nj):
        rest_inv = np.eye(4)
        rest_inv[:3, 3] = -joints[k]
        Gp.append(G[k] @ rest_inv)
    out = np.empty((nv, 3))
    for i in range(nv):
        t_p = t_s[i] + np.array([coeff @ posedirs[:, 3 * i + c] for c in range(3)])
        h = np.append(t_p, 1.0)
        acc = np.zeros(4)
        for k in range(nj):
            if weights[i, k] != 0.0:
                acc += weights[i, k] * (Gp[k] @ h)
        out[i] = acc[:3]
    return out, posed_joints
