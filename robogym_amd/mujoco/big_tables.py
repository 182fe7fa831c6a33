"""Static execution tables of the LARGE-MODEL stepper (`rb_step_kernel`, robogym_amd/csrc/rb_kernel.h), derived once on the
host from a compiled model: what `kernel_tables.py` is to the Shadow-hand kernel, for models beyond its compile-time layout
(dactyl/full_perpendicular: nv 168, 135 bodies, condim-6 contacts; reference
/root/reference/robogym/envs/dactyl/full_perpendicular.py:92-136, cube_env.py:239-242).

Everything here is a re-arrangement of `mjModel` index arrays (no physics):

* body levels (tree depth) for the top-down / bottom-up sweeps, subtree lists, the last dof of every body's chain
* the tree-sparse entry list of the joint-space inertia matrix M: (i, j) for every dof i and every ancestor dof j
* dof GROUPS: kinematic trees that a constraint can couple (static collision pairs, tendons) are merged; the Newton Hessian
  and M are factored group by group as dense blocks over the group's dofs
* the static collision pair list with mixed contact parameters (shared with kernel_tables: `collision_pairs`)
* static dof supports of the tendons, the constraint-row sources (friction-loss dofs / tendons, limited joints / tendons)
"""
import numpy as np

from robogym_amd.mujoco import mjcf_compiler as C
from robogym_amd.mujoco.kernel_tables import collision_pairs

TEN_W = 8        # dofs a tendon can depend on (RB_TENW)
CON_W = 24       # dofs a contact can depend on (RB_CONW)
MLONG = 8        # descendant lists longer than this are summed by a wave (RB_MLONG)
STAR_B = 5       # longest chain of a star tree (RB_STARB)


def _i32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int32))


def derive_big_tables(model):
    A = model.arrays
    nb, nv, ng = len(A["body_parentid"]), len(A["dof_bodyid"]), len(A["geom_type"])
    parent, dpar = A["body_parentid"], A["dof_parentid"]

    # ---- body levels (world = 0)
    depth = np.zeros(nb, dtype=int)
    for b in range(1, nb):
        depth[b] = depth[parent[b]] + 1
    lvl, adr = [], [0]
    for L in range(1, depth.max() + 1):
        lvl += [b for b in range(1, nb) if depth[b] == L]
        adr.append(len(lvl))
    A["b_lvl_body"], A["b_lvl_adr"] = _i32(lvl), _i32(adr)
    A["b_body_level"] = _i32(depth - 1)      # index of a body's level in the lists above (world: -1): the one-wave sweeps keep body b in lane b
    # ---- last dof of the chain a body hangs on (-1: none), dof chain lists root-first
    lastdof = np.full(nb, -1, dtype=np.int32)
    for b in range(1, nb):
        lastdof[b] = A["body_dofadr"][b] + A["body_dofnum"][b] - 1 if A["body_dofnum"][b] > 0 else lastdof[parent[b]]
    A["b_body_lastdof"] = lastdof
    chains = []
    for b in range(nb):
        c, i = [], int(lastdof[b])
        while i >= 0:
            c.append(i)
            i = int(dpar[i])
        chains.append(sorted(c))
    # ---- subtree lists (self included)
    sub = [[b] for b in range(nb)]
    for b in range(nb - 1, 0, -1):
        sub[parent[b]] += sub[b]
    sadr = [0]
    for b in range(nb):
        sadr.append(sadr[-1] + len(sub[b]))
    A["b_subtree_adr"], A["b_subtree"] = _i32(sadr), _i32([x for s in sub for x in sorted(s)])
    # ---- roots of the kinematic trees (MuJoCo's com-based frames use the subtree com of body_rootid)
    A["b_root_list"] = _i32(sorted(set(int(A["body_rootid"][b]) for b in range(1, nb))))
    # ---- M entries: (i, j) with j = i, parent(i), ...; Madr[i] = first entry of dof i
    Mi, Mj, Madr = [], [], []
    for i in range(nv):
        Madr.append(len(Mi))
        j = i
        while j >= 0:
            Mi.append(i); Mj.append(j)
            j = int(dpar[j])
    Madr.append(len(Mi))
    A["b_M_i"], A["b_M_j"], A["b_M_adr"] = _i32(Mi), _i32(Mj), _i32(Madr)
    # ---- trees and dof groups
    tree_of_dof = np.array([int(A["body_rootid"][A["dof_bodyid"][i]]) for i in range(nv)])
    trees = sorted(set(tree_of_dof.tolist()), key=lambda r: np.where(tree_of_dof == r)[0][0])
    for r in trees:   # a tree's dofs are contiguous (MuJoCo numbers dofs depth-first)
        idx = np.where(tree_of_dof == r)[0]
        assert (np.diff(idx) == 1).all()
    pairs, prm = collision_pairs(model)
    link = {r: r for r in trees}

    def find(r):
        while link[r] != r:
            r = link[r]
        return r

    def union(bodies):
        rs = [find(int(A["body_rootid"][b])) for b in bodies if A["body_weldid"][b] != 0 and lastdof[b] >= 0]
        for r in rs[1:]:
            link[find(r)] = find(rs[0])

    for a, b, _ in pairs:
        union([int(A["geom_bodyid"][a]), int(A["geom_bodyid"][b])])
    # tendon supports
    nt = len(A["tendon_adr"])
    ten_dofs = np.full((nt, TEN_W), -1, dtype=np.int32)
    for t in range(nt):
        adr_, num = int(A["tendon_adr"][t]), int(A["tendon_num"][t])
        dofs = set()
        if A["wrap_type"][adr_] == C.WRAP_JOINT:
            dofs = {int(A["jnt_dofadr"][A["wrap_objid"][w]]) for w in range(adr_, adr_ + num)}
            bodies = [int(A["jnt_bodyid"][A["wrap_objid"][w]]) for w in range(adr_, adr_ + num)]
        else:
            bodies = []
            for w in range(adr_, adr_ + num):
                ty = int(A["wrap_type"][w])
                if ty == C.WRAP_SITE:
                    bodies.append(int(A["site_bodyid"][A["wrap_objid"][w]]))
                elif ty in (C.WRAP_SPHERE, C.WRAP_CYLINDER):
                    bodies.append(int(A["geom_bodyid"][A["wrap_objid"][w]]))
            for x in bodies:   # any two bodies of the path: the dofs between them (symmetric difference of their chains)
                for y in bodies:
                    dofs |= set(chains[x]) ^ set(chains[y])
        if len(dofs) > TEN_W:
            raise NotImplementedError("tendon %d depends on %d dofs > %d" % (t, len(dofs), TEN_W))
        ten_dofs[t, :len(dofs)] = sorted(dofs)
        union(bodies)
    A["b_ten_dofs"] = ten_dofs
    # equality constraints couple trees too (weld: the two bodies; joint coupling: the two joints' bodies)
    eq_dofs = set()
    for e in range(len(A.get("eq_type", []))):
        o1, o2 = int(A["eq_obj1id"][e]), int(A["eq_obj2id"][e])
        if int(A["eq_type"][e]) == C.EQ_WELD:
            bodies = [o1, o2]
        else:
            bodies = [int(A["jnt_bodyid"][o1])] + ([int(A["jnt_bodyid"][o2])] if o2 >= 0 else [])
        union(bodies)
        for x in bodies:
            eq_dofs |= set(chains[x])
        w_eq = len(set().union(*[set(chains[x]) for x in bodies]))
        if w_eq > CON_W:
            raise NotImplementedError("an equality constraint depends on %d dofs > %d" % (w_eq, CON_W))
    groups = {}
    for r in trees:
        groups.setdefault(find(r), []).append(r)
    # a group = the dofs of its trees in ascending order (not necessarily one contiguous range: the target cube's tree sits between
    # the cube's and the hand's); b_dof_local = position of a dof inside its group = its row in the group's dense blocks
    gadr, gdofs, dof_group, dof_local = [0], [], np.zeros(nv, dtype=np.int32), np.zeros(nv, dtype=np.int32)
    order = sorted(groups.values(), key=lambda rs: min(np.where(tree_of_dof == r)[0][0] for r in rs))
    for gi, rs in enumerate(order):
        idx = np.sort(np.concatenate([np.where(tree_of_dof == r)[0] for r in rs]))
        dof_group[idx] = gi
        dof_local[idx] = np.arange(len(idx))
        gdofs += idx.tolist()
        gadr.append(len(gdofs))
    A["b_group_adr"], A["b_group_dofs"], A["b_dof_group"], A["b_dof_local"] = _i32(gadr), _i32(gdofs), dof_group, dof_local
    # ---- collision pairs
    A["b_pair_geom"] = _i32(pairs).reshape(-1, 3)
    A["b_pair_prm"] = np.ascontiguousarray(np.asarray(prm, dtype=np.float64).reshape(-1, 13)[:, :12])
    wmax = 0
    for a, b, _ in pairs:
        wmax = max(wmax, len(set(chains[A["geom_bodyid"][a]]) | set(chains[A["geom_bodyid"][b]])))
    if wmax > CON_W:
        raise NotImplementedError("a contact depends on %d dofs > %d" % (wmax, CON_W))
    # ---- constraint-row sources
    A["b_fric_dof"] = _i32([i for i in range(nv) if A["dof_frictionloss"][i] > 0])
    A["b_fric_ten"] = _i32([t for t in range(nt) if A["tendon_frictionloss"][t] > 0])
    A["b_lim_jnt"] = _i32([j for j in range(len(A["jnt_type"])) if A["jnt_limited"][j] and A["jnt_type"][j] in (C.JNT_HINGE, C.JNT_SLIDE)])
    A["b_lim_ten"] = _i32([t for t in range(nt) if A["tendon_limited"][t]])
    for j in range(len(A["jnt_type"])):
        if A["jnt_stiffness"][j] != 0 and A["jnt_type"][j] not in (C.JNT_HINGE, C.JNT_SLIDE):
            raise NotImplementedError("ball / free joint springs")
    # oriented bounding box of every geom in its own frame (centre, half extents): the broadphase's second test after the bounding spheres
    gt, gs = A["geom_type"], np.asarray(A["geom_size"], dtype=np.float64).reshape(-1, 3)
    mv_all = np.asarray(A["mesh_vert"], dtype=np.float64).reshape(-1, 3)
    aabb = np.zeros((len(gt), 6), dtype=np.float32)
    for g in range(len(gt)):
        t = int(gt[g])
        if t == C.GEOM_SPHERE:
            aabb[g, 3:] = gs[g, 0]
        elif t in (C.GEOM_CAPSULE, C.GEOM_CYLINDER):
            aabb[g, 3:] = (gs[g, 0], gs[g, 0], gs[g, 1] + (gs[g, 0] if t == C.GEOM_CAPSULE else 0.0))
        elif t in (C.GEOM_ELLIPSOID, C.GEOM_BOX):
            aabb[g, 3:] = gs[g]
        elif t == C.GEOM_MESH:
            mid = int(A["geom_dataid"][g])
            v = mv_all[int(A["mesh_vertadr"][mid]):int(A["mesh_vertadr"][mid]) + int(A["mesh_vertnum"][mid])]
            lo, hi = v.min(axis=0), v.max(axis=0)
            aabb[g, :3], aabb[g, 3:] = 0.5 * (lo + hi), 0.5 * (hi - lo)
        elif t == C.GEOM_PLANE:
            aabb[g, 3:] = 0.0
        else:
            raise NotImplementedError("geom type %d" % t)
    A["b_geom_aabb"] = aabb.reshape(-1)
    # mesh vertices as 16-byte records (x, y, z, vertex index): what the hull scans of the collision code read
    mv = np.asarray(A["mesh_vert"], dtype=np.float32).reshape(-1, 3)
    rec = np.zeros((len(mv), 4), dtype=np.float32)
    rec[:, :3] = mv
    for mesh in range(len(A["mesh_vertadr"])):
        a0, n = int(A["mesh_vertadr"][mesh]), int(A["mesh_vertnum"][mesh])
        rec[a0:a0 + n, 3] = np.arange(n, dtype=np.int32).view(np.float32)
    A["b_mesh_rec"] = rec.reshape(-1)
    # support cells of the hulls (kernel_tables.mesh_support_cells: per direction cell the vertices that can be the support point),
    # laid out as the hull scans read them: the first four records of a cell at a computable address, the rest in an overflow run
    from robogym_amd.mujoco.kernel_tables import CELL_N, mesh_support_cells

    cadr, vidx = mesh_support_cells(A["mesh_vertadr"], A["mesh_vertnum"], A["mesh_vert"])
    ncell = 6 * CELL_N * CELL_N
    nmesh = len(A["mesh_vertadr"])
    blk = np.zeros((nmesh * ncell, 4, 4), dtype=np.float32)
    ovf = [np.zeros(4, dtype=np.float32)]
    new_adr = np.zeros(nmesh * ncell, dtype=np.int32)
    for mi in range(nmesh):
        a0 = int(A["mesh_vertadr"][mi])
        for c in range(ncell):
            ce = mi * ncell + c
            e = int(cadr[ce]); start, cnt = e >> 8, e & 255
            new_adr[ce] = (len(ovf) << 8) | cnt
            for k in range(max(cnt, 4)):
                vi = int(vidx[start + min(k, cnt - 1)])
                r = np.zeros(4, dtype=np.float32)
                r[:3] = mv[a0 + vi]
                r[3] = np.array([vi], dtype=np.int32).view(np.float32)[0]
                if k < 4:
                    blk[ce, k] = r
                else:
                    ovf.append(r)
    A["b_cell_adr"], A["b_cell_blk"], A["b_cell_ovf"] = new_adr, blk.reshape(-1), np.concatenate(ovf)
    # every dof's entries of M below it: (entry index, descendant dof) pairs, for the owner-computes product M x
    dadr, dent, ddof = [0], [], []
    for i in range(nv):
        for e in range(len(Mi)):
            if Mj[e] == i and Mi[e] != i:
                dent.append(e); ddof.append(Mi[e])
        dadr.append(len(dent))
    A["b_Mdesc_adr"], A["b_Mdesc_ent"], A["b_Mdesc_dof"] = _i32(dadr), _i32(dent), _i32(ddof)
    # dofs whose descendant list is long (tree roots): rb_M_mul sums those lists with a whole wave; [count, dof, dof, ...]
    long_ = [i for i in range(nv) if dadr[i + 1] - dadr[i] > MLONG]
    A["b_Mlong"] = _i32([len(long_)] + long_)
    # friction-loss row of every dof (-1: none)
    fr = np.full(nv, -1, dtype=np.int32)
    for r, i in enumerate(A["b_fric_dof"]):
        fr[i] = r
    A["b_dof_fricrow"] = fr
    # ---- "star" trees: a chain of root dofs (each with exactly one child dof, up to 6) with simple chains of <= STAR_B dofs hanging off its
    # last dof (the cubes: 6 root dofs, 26 chains of 1 or 3 hinges; the hand: 2 wrist dofs, 5 finger chains of 4 or 5).  M and M + h B of such a
    # tree keep its sparsity: rb_star_solve eliminates the chains one thread each and the root block last instead of a dense factorisation.
    # A group whose trees are all stars solves with M / M + h B that way (b_star_grp[4 g + 3]); a group that is ONE star tree with no contact pair
    # and no tendon on it (the target cube) also has a tree-sparse Newton Hessian, M + a diagonal (b_star_grp[4 g]).
    # b_tree_desc[t] = (is star, first root dof, root dofs, first chain); b_tree_branch = (first dof, dofs) per chain; b_tree_adr[g] = first tree of group g
    touched = set()
    for a, b, _ in pairs:
        touched |= set(chains[A["geom_bodyid"][a]]) | set(chains[A["geom_bodyid"][b]])
    touched |= set(int(d) for d in ten_dofs.reshape(-1) if d >= 0)
    touched |= eq_dofs
    kids = [[] for _ in range(nv)]
    for i in range(nv):
        if dpar[i] >= 0:
            kids[int(dpar[i])].append(i)
    tdesc, tbr, tadr, sgrp = [], [], [0], []
    for gi in range(len(gadr) - 1):
        dofs = gdofs[gadr[gi]:gadr[gi + 1]]
        roots = sorted(set(tree_of_dof[dofs].tolist()), key=lambda r: np.where(tree_of_dof == r)[0][0])
        all_star = True
        for r in roots:
            td = np.where(tree_of_dof == r)[0].tolist()
            first = td[0]
            nroot, d = 1, first
            while len(kids[d]) == 1 and kids[d][0] == d + 1 and nroot < 6 and len(kids[d + 1]) != 0 and len(td) > nroot + 1:
                # (the chain goes on while the dof has exactly one child that itself has children: the last root dof is the one the chains hang off)
                d += 1
                nroot += 1
            root_last = first + nroot - 1
            star, br, i = True, [], root_last + 1
            while star and i <= td[-1]:
                if int(dpar[i]) != root_last:
                    star = False
                    break
                n_ = 1
                while i + n_ <= td[-1] and int(dpar[i + n_]) == i + n_ - 1 and len(kids[i + n_ - 1]) == 1:
                    n_ += 1
                if n_ > STAR_B or len(kids[i + n_ - 1]) != 0:
                    star = False
                br.append((i, n_))
                i += n_
            star = star and len(br) <= 128 and td == list(range(first, td[-1] + 1))
            tdesc.append([1 if star else 0, first, nroot, len(tbr)])
            tbr += br if star else []
            all_star = all_star and star
        tadr.append(len(tdesc))
        one = len(roots) == 1 and all_star and not (touched & set(dofs))
        sgrp.append([1 if one else 0, 0, 0, 1 if all_star else 0])
    A["b_star_grp"], A["b_tree_adr"], A["b_tree_desc"] = _i32(sgrp).reshape(-1), _i32(tadr), _i32(tdesc).reshape(-1)
    A["b_tree_branch"] = _i32(tbr + [(0, 0)]).reshape(-1)
    A["b_tree_brn_end"] = _i32([d_[3] for d_ in tdesc[1:]] + [len(tbr)])
    # trees of <= 8 dofs in one contiguous range, all of them (the rearrange worlds): rb_trees8_solve factors them side by side, eight lanes each.
    # [count, (first dof, dofs) ...]; count 0 = some tree does not qualify and the group-wise solves are used
    t8 = []
    for r in trees:
        td = np.where(tree_of_dof == r)[0].tolist()
        if len(td) > 8 or td != list(range(td[0], td[-1] + 1)):
            t8 = None
            break
        t8.append((td[0], len(td)))
    A["b_tree8"] = _i32([len(t8)] + [x for p_ in t8 for x in p_]) if t8 else _i32([0])
    A["b_dims"] = _i32([len(adr) - 1, len(Mi), len(pairs), len(gadr) - 1, max(np.diff(gadr)), len(A["b_root_list"]), wmax])
    return model
