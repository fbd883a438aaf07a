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


def lbs_vertex_subset(betas, pose, v_template, shapedirs, posedirs, j_regressor, parents, weights, vertex_ids):
    """The same statement for the listed vertices only (the joints still need every vertex: eq. 10, done as one float64
    product). Used as the float64 ARBITER of integer-pixel disagreements between two float32 evaluations
    (tests/test_gpu_parity_pixels.py): a few hundred (image, vertex) pairs out of millions."""
    betas, pose = np.asarray(betas, np.float64), np.asarray(pose, np.float64)
    v_template, shapedirs = np.asarray(v_template, np.float64), np.asarray(shapedirs, np.float64)
    posedirs, j_regressor, weights = np.asarray(posedirs, np.float64), np.asarray(j_regressor, np.float64), np.asarray(weights, np.float64)
    nj = j_regressor.shape[0]
    rot = [Rotation.from_rotvec(pose[k]).as_matrix() for k in range(nj)]
    coeff = np.concatenate([(rot[k] - np.eye(3)).reshape(9) for k in range(1, nj)])
    t_s = v_template + shapedirs @ betas                      # eq. 8/9, every vertex
    joints = j_regressor @ t_s                                # eq. 10
    G = [None] * nj
    for k in range(nj):
        local = np.eye(4)
        local[:3, :3] = rot[k]
        local[:3, 3] = joints[k] - (joints[parents[k]] if parents[k] >= 0 else 0.0)
        G[k] = local if parents[k] < 0 else G[parents[k]] @ local
    Gp = []
    for k in range(nj):
        rest_inv = np.eye(4)
        rest_inv[:3, 3] = -joints[k]
        Gp.append(G[k] @ rest_inv)
    out = np.empty((len(vertex_ids), 3))
    for n, i in enumerate(vertex_ids):
        t_p = t_s[i] + coeff @ posedirs[:, 3 * i:3 * i + 3]
        h = np.append(t_p, 1.0)
        out[n] = sum(weights[i, k] * (Gp[k] @ h) for k in range(nj) if weights[i, k] != 0.0)[:3]
    return out


def projected_pixels_subset(params_row, vertex_ids, v_template, shapedirs, posedirs, j_regressor, parents, weights,
                            image_size: float = 256.0, mesh_offset_z: float = 0.05):
    """float64 pixel coordinates [n, 3] of the listed vertices of ONE image, from its float32 params row (dad_3dnet.yaml layout:
    300 shape | 100 expression | 3 jaw | 6 rotation | 3 translation | 1 scale): lbs above, then model_training/model/flame.py:224-228
    (z += 0.05, 6-DoF rotation: model/utils.py:92-101 with F.normalize's eps 1e-12) and head_mesh.py:39-45 (scale clamp, tz := 0,
    (v + 1) / 2 * image_size), every operation in float64."""
    p = np.asarray(params_row, np.float64)
    pose = np.zeros((5, 3))
    pose[2] = p[400:403]
    v = lbs_vertex_subset(p[:400], pose, v_template, shapedirs, posedirs, j_regressor, parents, weights, vertex_ids)
    v[:, 2] += mesh_offset_z
    unit = lambda x: x / max(np.linalg.norm(x), 1e-12)  # noqa: E731
    b1 = unit(p[403:406])
    b3 = unit(np.cross(b1, p[406:409]))
    b2 = -np.cross(b1, b3)
    R = np.stack((b1, b2, b3), axis=-1)
    v = v @ R.T
    s = max(p[412] + 1.0, 1e-8)
    v = v * s + np.array([p[409], p[410], 0.0])
    return (v + 1.0) / 2.0 * image_size
