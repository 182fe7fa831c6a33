"""Static tables the batched HIP stepper needs on top of the mjModel-style arrays.

Everything here is derived once per model on the host and appended to the model
arrays (prefix ``k_``) before the blob crosses the C ABI.  The tables encode the
MI355X execution plan (one 64-lane wavefront per env):

* level-ordered body / dof lists, so tree recursions become a short sequence of
  lane-parallel sweeps separated by wave barriers;
* per-body dof-chain bit masks and per-dof descendant lists, so sparse Jacobian
  rows and the tree-sparse L'DL factorisation are gathers (deterministic, no
  LDS atomics);
* the static candidate list of collidable geom pairs with their mixed contact
  parameters (the body-pair filters of the collision driver — same weld group,
  parent/child, explicit excludes, contype/conaffinity — depend on the model only);
* per-dof gather lists for tendon and actuator force scatter.
"""
import numpy as np

from robogym_amd.mujoco import mjcf_compiler as C


def _i32(x):
    return np.asarray(x, dtype=np.int32)


def collision_pairs(model):
    """The static candidate list of the collision driver (engine_collision_driver.c filters: same weld group, parent-child,
    explicit excludes, contype / conaffinity) with each pair's mixed contact parameters (mj_contactParam: margin / gap max,
    friction element-wise max, solref / solimp by solmix, condim max).  Returns ([(g1, g2, condim)], [12 floats per pair:
    margin, gap, friction3, solref2, solimp5]); g1 has the smaller geom type (planes first)."""
    A = model.arrays
    nbody, ngeom = len(A["body_parentid"]), len(A["geom_type"])
    parent = A["body_parentid"]
    gtype, gbody = A["geom_type"], A["geom_bodyid"]
    weld = A["body_weldid"]
    excl = set(int(s) for s in A["exclude_signature"])
    geoms_of = [[g for g in range(ngeom) if gbody[g] == b] for b in range(nbody)]
    pairs, prm = [], []
    for b1 in range(nbody):
        if not geoms_of[b1]:
            continue
        for b2 in range(b1 + 1, nbody):
            if not geoms_of[b2]:
                continue
            w1, w2 = weld[b1], weld[b2]
            if w1 == w2:
                continue
            pw1, pw2 = weld[parent[w1]], weld[parent[w2]]
            if w1 != 0 and w2 != 0 and (w1 == pw2 or w2 == pw1):
                continue
            if ((b1 << 16) + b2) in excl:
                continue
            for g1 in geoms_of[b1]:
                for g2 in geoms_of[b2]:
                    a, b = (g1, g2) if gtype[g1] <= gtype[g2] else (g2, g1)
                    if not ((A["geom_contype"][a] & A["geom_conaffinity"][b]) or (A["geom_contype"][b] & A["geom_conaffinity"][a])):
                        continue
                    if gtype[a] == C.GEOM_PLANE and gtype[b] == C.GEOM_PLANE:
                        continue
                    margin = max(A["geom_margin"][a], A["geom_margin"][b])
                    gap = max(A["geom_gap"][a], A["geom_gap"][b])
                    fr = np.maximum(A["geom_friction"][a], A["geom_friction"][b])
                    m1, m2 = A["geom_solmix"][a], A["geom_solmix"][b]
                    if m1 >= 1e-15 and m2 >= 1e-15:
                        mix = m1 / (m1 + m2)
                    elif m1 < 1e-15 and m2 < 1e-15:
                        mix = 0.5
                    else:
                        mix = 0.0 if m1 < 1e-15 else 1.0
                    r1, r2 = A["geom_solref"][a], A["geom_solref"][b]
                    solref = mix * r1 + (1 - mix) * r2 if (r1[0] > 0 and r2[0] > 0) else np.minimum(r1, r2)
                    solimp = mix * A["geom_solimp"][a] + (1 - mix) * A["geom_solimp"][b]
                    condim = max(A["geom_condim"][a], A["geom_condim"][b])
                    pairs.append((a, b, condim))
                    prm.append(np.concatenate([[margin, gap], fr, solref, solimp, [mix]]))  # 12 floats + the solmix weight of geom a
    return pairs, prm


def derive_kernel_tables(model, max_row_nnz=16):
    A = model.arrays
    nbody, nv, ngeom = len(A["body_parentid"]), len(A["dof_bodyid"]), len(A["geom_type"])
    parent = A["body_parentid"]
    if nv > 64:
        raise NotImplementedError("dof-chain masks are 64-bit: nv=%d" % nv)

    # ---------------------------------------------------------------- bodies: static vs dynamic, levels
    from robogym_amd.mujoco.setconst import kinematics

    kin0 = kinematics(model, A["qpos0"])
    is_static = np.array([A["body_weldid"][b] == 0 for b in range(nbody)])
    A["k_body_static"] = _i32(is_static)
    A["k_static_xpos"] = kin0["xpos"].copy()
    A["k_static_xquat"] = kin0["xquat"].copy()
    depth = np.zeros(nbody, dtype=int)
    for b in range(1, nbody):
        p = parent[b]
        depth[b] = 0 if is_static[b] else (depth[p] + 1 if not is_static[p] else 1)
    dyn = [b for b in range(1, nbody) if not is_static[b]]
    nlevel = max([depth[b] for b in dyn], default=0)
    lvl, adr = [], [0]
    for L in range(1, nlevel + 1):
        lvl += [b for b in dyn if depth[b] == L]
        adr.append(len(lvl))
    A["k_lvl_body"] = _i32(lvl)
    A["k_lvl_body_adr"] = _i32(adr)
    A["k_static_body"] = _i32([b for b in range(nbody) if is_static[b]])

    # ---------------------------------------------------------------- com-frame origins per kinematic tree
    # single-body trees use the body's own com (exactly subtree_com); trees hanging off a static
    # root use the constant subtree com at qpos0.  Any origin gives the same M / bias analytically.
    root = A["body_rootid"]
    origin_body = np.full(nbody, -1, dtype=np.int32)
    origin_const = np.zeros((nbody, 3))
    for r in set(int(x) for x in root[1:]):
        members = [b for b in range(1, nbody) if root[b] == r]
        if len(members) == 1 and not is_static[r]:
            origin_body[r] = r
        else:
            mass = A["body_mass"][members]
            origin_const[r] = (kin0["xipos"][members] * mass[:, None]).sum(0) / max(mass.sum(), 1e-12)
            if not is_static[r]:
                # moving multi-body tree: fall back to the root body's com as origin
                origin_body[r] = r
    # origin slots: one per kinematic tree that owns dofs; dof-less trees (floor, world) share slot 0
    roots_with_dofs = sorted(set(int(root[A["dof_bodyid"][d]]) for d in range(nv)))
    if len(roots_with_dofs) > 3:
        raise NotImplementedError("more than 3 kinematic trees with dofs (origin slots)")
    A["k_body_orgslot"] = _i32([1 + roots_with_dofs.index(int(root[b])) if int(root[b]) in roots_with_dofs else 0 for b in range(nbody)])
    A["k_root_origin_body"] = origin_body
    A["k_root_origin_const"] = origin_const

    # ---------------------------------------------------------------- dof chains
    dpar = A["dof_parentid"]
    mask = np.zeros(nbody, dtype=np.uint64)
    body_lastdof = np.full(nbody, -1, dtype=np.int32)
    for b in range(1, nbody):
        bb = b
        while bb > 0 and A["body_dofnum"][bb] == 0:
            bb = parent[bb]
        if bb > 0:
            i = A["body_dofadr"][bb] + A["body_dofnum"][bb] - 1
            body_lastdof[b] = i
            while i >= 0:
                mask[b] |= np.uint64(1) << np.uint64(i)
                i = dpar[i]
    A["k_body_dofmask"] = mask.view(np.int32).reshape(nbody, 2).copy()
    A["k_body_lastdof"] = body_lastdof
    ddepth = np.zeros(nv, dtype=int)
    for i in range(nv):
        ddepth[i] = 0 if dpar[i] < 0 else ddepth[dpar[i]] + 1
    ndl = ddepth.max() + 1 if nv else 0
    dl, dadr = [], [0]
    for L in range(ndl):
        dl += [i for i in range(nv) if ddepth[i] == L]
        dadr.append(len(dl))
    A["k_lvl_dof"] = _i32(dl)
    A["k_lvl_dof_adr"] = _i32(dadr)
    # sparse inertia entries (i, j) with j an ancestor-or-self of i, ordered by level of i (deepest last)
    Mi, Mj, Madr = [], [], [0]
    for L in range(ndl):
        for i in [x for x in range(nv) if ddepth[x] == L]:
            j = i
            while j >= 0:
                Mi.append(i); Mj.append(j)
                j = dpar[j]
        Madr.append(len(Mi))
    A["k_M_i"] = _i32(Mi); A["k_M_j"] = _i32(Mj); A["k_M_lvl_adr"] = _i32(Madr)
    # descendants of each dof (strict)
    desc = [[] for _ in range(nv)]
    for k in range(nv):
        j = dpar[k]
        while j >= 0:
            desc[j].append(k)
            j = dpar[j]
    dadr2, dflat = [0], []
    for i in range(nv):
        dflat += desc[i]
        dadr2.append(len(dflat))
    A["k_desc_adr"] = _i32(dadr2); A["k_desc"] = _i32(dflat)

    # subtree membership (self included) for composite-inertia / force gathers
    sadr, sflat = [0], []
    for b in range(nbody):
        members = []
        for c in range(nbody):
            a = c
            while a > 0 and a != b:
                a = parent[a]
            if a == b and (b > 0 or c == 0):
                members.append(c)
        sflat += members
        sadr.append(len(sflat))
    A["k_subtree_adr"] = _i32(sadr); A["k_subtree"] = _i32(sflat)
    if nbody > 32:
        raise NotImplementedError("subtree masks are 32-bit: nbody=%d" % nbody)
    # the same membership as one bit mask per body (bit c = body c is in the subtree of b): the gathers iterate set bits
    # in ascending order instead of chasing the index list through global memory
    A["k_subtree_mask"] = np.array([sum(1 << int(c) for c in sflat[sadr[b]:sadr[b + 1]]) for b in range(nbody)], dtype=np.uint32).view(np.int32)
    # dofs whose motion precedes dof d on its chain (velocity "before" the joint, mj_comVel):
    # strict ancestors, minus the sibling rotational dofs of the same ball / free joint
    velmask = np.zeros(nv, dtype=np.uint64)
    for d in range(nv):
        j = A["dof_jntid"][d]
        t, da = A["jnt_type"][j], A["jnt_dofadr"][j]
        a = dpar[d]
        while a >= 0:
            sibling = False
            if t == C.JNT_BALL and a >= da:
                sibling = True
            if t == C.JNT_FREE and a >= da + 3:
                sibling = True
            if not sibling:
                velmask[d] |= np.uint64(1) << np.uint64(a)
            a = dpar[a]
    A["k_dof_velmask"] = velmask.view(np.int32).reshape(nv, 2).copy()

    # ---------------------------------------------------------------- collision pair list
    gtype, gbody = A["geom_type"], A["geom_bodyid"]
    pairs, prm = collision_pairs(model)
    max_nnz = max([bin(int(mask[gbody[a]] | mask[gbody[b]])).count("1") for a, b, _ in pairs], default=0)
    if max_nnz > max_row_nnz:
        raise NotImplementedError("contact row needs %d nonzeros > %d" % (max_nnz, max_row_nnz))
    A["k_pair_geom"] = _i32(pairs).reshape(-1, 3)
    # geoms whose size follows the env's `geom_scale` parameter: what RandomizedCubeSizeWrapper rescales (wrappers/cube.py:12-53)
    gnames = model.names["geom"]
    A["k_geom_scaled"] = _i32([1 if (n in ("cube:middle", "cube:top", "cube:bottom") and int(A["geom_type"][g]) == 6) else 0 for g, n in enumerate(gnames)])
    prm = np.asarray(prm, dtype=np.float64).reshape(-1, 13)
    A["k_pair_prm"] = np.ascontiguousarray(prm[:, :12])
    A["k_pair_mix"] = np.ascontiguousarray(prm[:, 12])
    # oriented bounding boxes in the geom frame (conservative pre-filter before MPR)
    aabb = np.zeros((ngeom, 3))
    for g in range(ngeom):
        t, s = gtype[g], A["geom_size"][g]
        if t in (C.GEOM_BOX, C.GEOM_MESH, C.GEOM_ELLIPSOID):
            aabb[g] = s
        elif t == C.GEOM_SPHERE:
            aabb[g] = s[0]
        elif t in (C.GEOM_CAPSULE,):
            aabb[g] = [s[0], s[0], s[0] + s[1]]
        elif t == C.GEOM_CYLINDER:
            aabb[g] = [s[0], s[0], s[1]]
    A["k_geom_aabb"] = aabb

    # ---------------------------------------------------------------- hull support tables per direction cell
    A["k_mesh_cell_adr"], A["k_mesh_cell_vidx"] = mesh_support_cells(A["mesh_vertadr"], A["mesh_vertnum"], A["mesh_vert"])

    # ---------------------------------------------------------------- tendon static dof supports
    nt = len(A["tendon_adr"])
    tdofs = np.full((nt, 4), -1, dtype=np.int32)
    for t in range(nt):
        adr, num = A["tendon_adr"][t], A["tendon_num"][t]
        support = set()
        if A["wrap_type"][adr] == C.WRAP_JOINT:
            for w in range(adr, adr + num):
                support.add(int(A["jnt_dofadr"][A["wrap_objid"][w]]))
        else:
            bodies = []
            for w in range(adr, adr + num):
                wt = A["wrap_type"][w]
                if wt == C.WRAP_SITE:
                    bodies.append(int(A["site_bodyid"][A["wrap_objid"][w]]))
                elif wt in (C.WRAP_SPHERE, C.WRAP_CYLINDER):
                    bodies.append(int(A["geom_bodyid"][A["wrap_objid"][w]]))
            # straight segments can also skip the wrap geom: consider all body pairs along the path
            for x in range(len(bodies)):
                for y in range(x + 1, min(x + 3, len(bodies))):
                    sym = int(mask[bodies[x]] ^ mask[bodies[y]])
                    support |= {i for i in range(nv) if (sym >> i) & 1}
        support = sorted(support)
        if len(support) > 4:
            raise NotImplementedError("tendon %d touches %d dofs (>4)" % (t, len(support)))
        tdofs[t, : len(support)] = support
    A["k_ten_dofs"] = tdofs
    # per tendon: its path as a list of 8-word records, so that the tendon stage reads ONE record per stretch instead of walking
    # wrap_type -> wrap_objid -> site_bodyid -> body_dofmask chains of dependent loads:
    #   joint term      : [0, qposadr, slot of the dof in k_ten_dofs, coef]
    #   site -> site    : [1, s0 | s1 << 8 | 255 << 16 | 255 << 24, ba | bb << 8, 1 / divisor, 0, chain bits]
    #   site -> (wrap geom) -> site : [2 | geom wrap type << 4, s0 | s1 << 8 | g << 16 | sidesite << 24, ba | bb << 8 | bg << 16, 1 / divisor, radius, chain bits]
    # chain bits: bit 3 e + k = "dof k_ten_dofs[t][e] moves body k" (k = 0: ba, 1: bb, 2: bg).  Floats are stored by bit pattern.
    f2i = lambda x: int(np.float32(x).view(np.int32))
    recs, radr = [], [0]
    for t in range(nt):
        adr, num = int(A["tendon_adr"][t]), int(A["tendon_num"][t])
        td = [int(d) for d in tdofs[t]]
        if A["wrap_type"][adr] == C.WRAP_JOINT:
            for w in range(adr, adr + num):
                j = int(A["wrap_objid"][w]); d = int(A["jnt_dofadr"][j])
                recs.append([0, int(A["jnt_qposadr"][j]), td.index(d), f2i(A["wrap_prm"][w]), 0, 0, 0, 0])
        else:
            divisor, w = 1.0, adr
            while w < adr + num - 1:
                t0, t1 = int(A["wrap_type"][w]), int(A["wrap_type"][w + 1])
                if t0 == C.WRAP_PULLEY or t1 == C.WRAP_PULLEY:
                    if t0 == C.WRAP_PULLEY:
                        divisor = float(A["wrap_prm"][w])
                    w += 1
                    continue
                s0 = int(A["wrap_objid"][w]); ba = int(A["site_bodyid"][s0])
                if t1 in (C.WRAP_SPHERE, C.WRAP_CYLINDER):
                    g, s1, sid = int(A["wrap_objid"][w + 1]), int(A["wrap_objid"][w + 2]), int(A["wrap_prm"][w + 1])
                    kind, bg, radius = 2 | (t1 << 4), int(A["geom_bodyid"][g]), float(A["geom_size"][g][0])
                    w += 2
                else:
                    g, s1, sid, kind, bg, radius = 255, int(A["wrap_objid"][w + 1]), -1, 1, 0, 0.0
                    w += 1
                bb = int(A["site_bodyid"][s1])
                bits = 0
                for e, d in enumerate(td):
                    if d >= 0:
                        for k, b in enumerate((ba, bb, bg)):
                            if (int(mask[b]) >> d) & 1:
                                bits |= 1 << (3 * e + k)
                if max(s0, s1, g if g != 255 else 0, max(sid, 0)) > 254 or max(ba, bb, bg) > 255:
                    raise NotImplementedError("tendon path record: index beyond 8 bits")
                recs.append([kind, s0 | (s1 << 8) | (g << 16) | ((sid if sid >= 0 else 255) << 24), ba | (bb << 8) | (bg << 16), f2i(1.0 / divisor), f2i(radius), bits, 0, 0])
        radr.append(len(recs))
    A["k_ten_path"] = np.array(recs, dtype=np.int64).astype(np.uint32).view(np.int32).reshape(-1, 8) if recs else np.zeros((0, 8), np.int32)
    A["k_ten_path_adr"] = _i32(radr)
    # per-dof gather lists: (tendon, slot)
    adr, flat = [0], []
    for i in range(nv):
        for t in range(nt):
            for s in range(4):
                if tdofs[t, s] == i:
                    flat.append((t, s))
        adr.append(len(flat))
    A["k_dof_ten_adr"] = _i32(adr); A["k_dof_ten"] = _i32(flat).reshape(-1, 2)
    # per-dof actuator gather lists: (actuator, tendon slot or -1 for joint transmission)
    adr, flat = [0], []
    for i in range(nv):
        for u in range(len(A["actuator_trntype"])):
            if A["actuator_trntype"][u] == C.TRN_JOINT:
                if A["jnt_dofadr"][A["actuator_trnid"][u]] == i:
                    flat.append((u, -1))
            else:
                t = A["actuator_trnid"][u]
                for s in range(4):
                    if tdofs[t, s] == i:
                        flat.append((u, s))
        adr.append(len(flat))
    A["k_dof_act_adr"] = _i32(adr); A["k_dof_act"] = _i32(flat).reshape(-1, 2)

    # ---------------------------------------------------------------- constraint row sources
    A["k_fric_dof"] = _i32([i for i in range(nv) if A["dof_frictionloss"][i] > 0])
    A["k_fric_ten"] = _i32([t for t in range(nt) if A["tendon_frictionloss"][t] > 0])
    A["k_lim_jnt"] = _i32([j for j in range(len(A["jnt_type"])) if A["jnt_limited"][j] and A["jnt_type"][j] in (C.JNT_SLIDE, C.JNT_HINGE)])
    A["k_lim_ten"] = _i32([t for t in range(nt) if A["tendon_limited"][t]])
    # ---------------------------------------------------------------- kinematic trees as dense inertia blocks
    # dofs of one tree are contiguous; M (and M + hB) is block diagonal over trees.  A tree is "constrained"
    # when any constraint row can ever touch it (friction loss, limits, tendons, collidable geoms); the
    # Newton solve runs in the compact space of constrained dofs, the others keep qacc = qacc_smooth.
    dof_root = [int(root[A["dof_bodyid"][d]]) for d in range(nv)]
    trees = []
    for d in range(nv):
        if not trees or trees[-1][0] != dof_root[d]:
            trees.append([dof_root[d], d, 0])
        trees[-1][2] += 1
    assert len(set(t[0] for t in trees)) == len(trees), "dofs of a kinematic tree must be contiguous"
    col_bodies = set()
    for a, b, _ in pairs:
        col_bodies.add(int(gbody[a])); col_bodies.add(int(gbody[b]))
    constrained_dof = np.zeros(nv, dtype=bool)
    constrained_dof[A["dof_frictionloss"] > 0] = True
    for j in A["k_lim_jnt"] if "k_lim_jnt" in A else []:
        constrained_dof[A["jnt_dofadr"][j]] = True
    for j in range(len(A["jnt_type"])):
        if A["jnt_limited"][j] and A["jnt_type"][j] in (C.JNT_SLIDE, C.JNT_HINGE):
            constrained_dof[A["jnt_dofadr"][j]] = True
    for t in range(nt):
        for e in range(4):
            if tdofs[t, e] >= 0:
                constrained_dof[tdofs[t, e]] = True
    for b in col_bodies:
        for d in range(nv):
            if (int(mask[b]) >> d) & 1:
                constrained_dof[d] = True
    madr, blk_words = [], 0
    dof_blk = np.zeros(nv, dtype=np.int32)
    dof_blk2 = np.zeros(nv, dtype=np.int32)   # row stride | block start word << 8
    tree_tab = []
    d2c = np.full(nv, -1, dtype=np.int32)
    c2d = []
    for (r, s0, n) in trees:
        stride = (n + 3) // 4 * 4
        if stride % 8 == 0:
            stride += 4          # conflict-free 16-byte row reads across lanes
        cons = bool(constrained_dof[s0:s0 + n].any())
        cbase = len(c2d) if cons else 255
        if cons:
            for d in range(s0, s0 + n):
                d2c[d] = len(c2d); c2d.append(d)
        for d in range(s0, s0 + n):
            mrow = blk_words + (d - s0) * stride
            assert mrow < 65536 and s0 < 256 and n < 256
            dof_blk[d] = mrow | (s0 << 16) | (n << 24)
            dof_blk2[d] = stride | (blk_words << 8)
        tree_tab.append([s0, n, blk_words, stride, cbase])
        blk_words += n * stride
    nvc = len(c2d)
    hs = (nvc + 3) // 4 * 4
    if hs % 8 == 0:
        hs += 4
    c_blk = np.zeros(max(nvc, 1), dtype=np.int32)
    for i, d in enumerate(c2d):
        s0, n = (dof_blk[d] >> 16) & 255, (dof_blk[d] >> 24) & 255
        c_blk[i] = (dof_blk[d] & 0xFFFF) | (int(d2c[s0]) << 16) | (n << 24)
    A["k_dof_blk"] = dof_blk; A["k_dof_blk2"] = dof_blk2; A["k_c_blk"] = c_blk
    A["k_d2c"] = d2c; A["k_c2d"] = _i32(c2d)
    A["k_tree"] = _i32(tree_tab).reshape(-1, 5)
    A["k_blk_dims"] = _i32([nvc, hs, blk_words, len(trees), max(t[2] for t in trees)])

    # ---------------------------------------------------------------- tree-sparse L'DL of M (mj_factorM / mj_solveM)
    # M = L' D L over the dof tree touches only (dof, ancestor) entries.  Dofs of equal depth are independent, so the
    # factorisation is one lane-parallel pass per depth, deepest first: for a dof k and ancestors j <= i < k,
    # M[i][j] -= M[k][i] M[k][j] / M[k][k].  The passes are stored as rounds of 64 descriptors (padded with zeros) so
    # that a lane prefetches all of its descriptors with one batch of loads.
    #   triple: w0 = addr(k,i) | addr(k,j) << 10 | addr(k,k) << 20, w1 = addr(i,j) | valid << 31
    #   pair  : k | i << 6 | addr(k,i) << 12 | addr(k,k) << 22 (all-ones: padding)
    # addr = word inside the per-tree block storage (lower triangle: row >= column).
    dpar = A["dof_parentid"]
    depth = np.zeros(nv, dtype=int)
    for k in range(nv):
        depth[k] = 0 if dpar[k] < 0 else depth[dpar[k]] + 1

    def addr(a, b):
        assert a >= b
        return int(dof_blk[a] & 0xFFFF) + (b - int((dof_blk[a] >> 16) & 255))

    assert blk_words < 1024
    def ltdl_rounds(dofs, index_of):
        """descriptor rounds for the trees spanned by `dofs`; pair fields k, i are written as index_of[dof]"""
        tri_rounds, pair_rounds = [], []
        for lev in range(int(depth.max()), 0, -1):
            tri, pr = [], []
            for k in dofs:
                if depth[k] != lev:
                    continue
                anc, i = [], dpar[k]
                while i >= 0:
                    anc.append(int(i)); i = dpar[i]
                for i in anc:
                    pr.append(int(index_of[k]) | (int(index_of[i]) << 6) | (addr(k, i) << 12) | (addr(k, k) << 22))
                    for j in anc:
                        if j <= i:
                            tri.append((addr(k, i) | (addr(k, j) << 10) | (addr(k, k) << 20), addr(i, j) | (1 << 31)))
            for lst, rounds, pad in ((tri, tri_rounds, (0, 0)), (pr, pair_rounds, -1)):
                for r0 in range(0, len(lst), 64):
                    chunk = lst[r0:r0 + 64]
                    rounds.append(chunk + [pad] * (64 - len(chunk)))
        t = np.array(tri_rounds, dtype=np.int64).astype(np.uint32).view(np.int32).reshape(-1, 2) if tri_rounds else np.zeros((0, 2), np.int32)
        q = np.array(pair_rounds, dtype=np.int64).astype(np.uint32).view(np.int32).reshape(-1) if pair_rounds else np.zeros(0, np.int32)
        return t, q

    A["k_ltdl_tri"], A["k_ltdl_pair"] = ltdl_rounds(list(range(nv)), list(range(nv)))
    # the same passes for the Newton Hessian when it has the tree pattern (every contact couples one dof chain only):
    # constrained trees only, vectors indexed by compact dof
    A["k_ltdl_tri_c"], A["k_ltdl_pair_c"] = ltdl_rounds([d for d in range(nv) if d2c[d] >= 0], d2c)
    # static precondition: every tendon's dof support lies on one chain (its J'DJ block then fits the tree pattern)
    chain_ok = 1
    for t in range(len(A["k_ten_dofs"])):
        ds = sorted(int(d) for d in A["k_ten_dofs"][t] if d >= 0)
        for a_ in ds:
            for b_ in ds:
                if a_ > b_:
                    i = a_
                    while i >= 0 and i != b_:
                        i = dpar[i]
                    if i != b_:
                        chain_ok = 0
    A["k_tree_newton_ok"] = _i32([chain_ok])
    A["k_dims"] = _i32([nlevel, ndl, len(Mi), len(pairs), len(A["k_static_body"]), max_nnz])
    return model


# ------------------------------------------------------------------------------------ convex-hull support cells
CELL_N = 8                      # RG_CELLN in rg_types.h: cube-map grid per face, 6 * 8 * 8 = 384 direction cells
_cell_cache = {}


def _direction_cells(n=CELL_N):
    """(centre, angular radius) of every direction cell.  Cell id = (face * n + iu) * n + iv with
    face = 2 * axis + (direction[axis] < 0), u / v the (axis+1)%3 / (axis+2)%3 components divided by |major|
    — the same arithmetic as dir_cell() in rg_kernel.h."""
    out = []
    for f in range(6):
        ax, sg = f // 2, (1.0 if f % 2 == 0 else -1.0)
        for iu in range(n):
            for iv in range(n):
                corners = []
                for du in (0, 1):
                    for dv in (0, 1):
                        x = np.zeros(3)
                        x[ax], x[(ax + 1) % 3], x[(ax + 2) % 3] = sg, -1 + 2 * (iu + du) / n, -1 + 2 * (iv + dv) / n
                        corners.append(x / np.linalg.norm(x))
                corners = np.array(corners)
                c = corners.mean(0)
                c /= np.linalg.norm(c)
                rho = np.arccos(np.clip(corners @ c, -1, 1)).max() + 2e-3  # slack: fp32 cell assignment at the borders
                out.append((c, rho))
    return out


def mesh_support_cells(vertadr, vertnum, vert):
    """For every mesh and direction cell: the hull vertices that can be the support point for SOME direction
    of the cell (ascending vertex index).  A vertex v is dropped only when a single other vertex u beats it
    over the whole cap of the cell by a clear margin ((u - v).d > 1e-5 * hull size for every d in the cap), so
    the arg-max over the list equals the arg-max over the full hull bit for bit, ties included.
    Returns (adr [nmesh * ncell] = first record << 8 | count, vidx [nrec] = vertex index inside the mesh)."""
    vert = np.asarray(vert, dtype=np.float64).reshape(-1, 3)
    key = (vert.tobytes(), np.asarray(vertadr).tobytes(), np.asarray(vertnum).tobytes())
    if key in _cell_cache:
        return _cell_cache[key]
    cells = _direction_cells()
    adr, vidx = [], []
    for mi in range(len(vertnum)):
        P = vert[int(vertadr[mi]): int(vertadr[mi]) + int(vertnum[mi])]
        dn = np.linalg.norm(P[None, :, :] - P[:, None, :], axis=2)   # dn[v, u] = |u - v|
        tol = 1e-5 * np.abs(P).max()
        for c, rho in cells:
            pc = P @ c
            dominated = ((pc[None, :] - pc[:, None]) - dn * np.sin(rho) > tol).any(axis=1)
            keep = np.nonzero(~dominated)[0]
            if len(keep) > 255:
                raise NotImplementedError("support cell with %d candidate vertices" % len(keep))
            adr.append((len(vidx) << 8) | len(keep))
            vidx.extend(int(k) for k in keep)
    out = (_i32(adr), _i32(vidx))
    _cell_cache[key] = out
    return out
