"""Model constants that depend on the reference configuration qpos0
(the role of MuJoCo's `mj_setConst`, which the reference reaches through
`SimulationInterface.set_constants`, /root/reference/robogym/mujoco/simulation_interface.py:199-201).

Computed once on the host in double precision with plain numpy:
subtree masses, the joint-space inertia matrix at qpos0, and from it the
inverse-weight tables (`body_invweight0`, `dof_invweight0`, `tendon_invweight0`),
`tendon_length0`, `actuator_acc0` and `stat_meaninertia`, which parameterise
the constraint regulariser R and the solver's stopping scale.

The small kinematics / Jacobian / tendon-path routines here are host utilities
of the compiler; the per-step versions live in the HIP kernels.
"""
import numpy as np

from robogym_amd.mujoco import mjcf_compiler as C

MINVAL = 1e-15


def kinematics(m, qpos):
    """World poses of bodies, joint anchors/axes, geoms and sites for one configuration."""
    A = m.arrays
    nbody = len(A["body_parentid"])
    xpos = np.zeros((nbody, 3)); xquat = np.zeros((nbody, 4)); xquat[0, 0] = 1
    xmat = np.zeros((nbody, 3, 3)); xmat[0] = np.eye(3)
    njnt = len(A["jnt_type"])
    xanchor = np.zeros((njnt, 3)); xaxis = np.zeros((njnt, 3))
    for b in range(1, nbody):
        p = A["body_parentid"][b]
        pos = xpos[p] + xmat[p] @ A["body_pos"][b]
        quat = C.qmul(xquat[p], A["body_quat"][b])
        for j in range(A["body_jntadr"][b], A["body_jntadr"][b] + A["body_jntnum"][b]) if A["body_jntnum"][b] else []:
            t, qa = A["jnt_type"][j], A["jnt_qposadr"][j]
            if t == C.JNT_FREE:
                pos = qpos[qa:qa + 3].copy(); quat = C.qnorm(qpos[qa + 3:qa + 7])
                xanchor[j] = pos; xaxis[j] = [0, 0, 1]
                continue
            R = C.q2mat(quat)
            xanchor[j] = pos + R @ A["jnt_pos"][j]
            xaxis[j] = R @ A["jnt_axis"][j]
            if t == C.JNT_SLIDE:
                pos = pos + xaxis[j] * (qpos[qa] - A["qpos0"][qa])
            elif t == C.JNT_BALL:
                quat = C.qmul(quat, C.qnorm(qpos[qa:qa + 4]))
                pos = xanchor[j] - C.q2mat(quat) @ A["jnt_pos"][j]
            else:
                quat = C.qmul(quat, C.axisangle2q(A["jnt_axis"][j], qpos[qa] - A["qpos0"][qa]))
                pos = xanchor[j] - C.q2mat(quat) @ A["jnt_pos"][j]
        xpos[b] = pos; xquat[b] = C.qnorm(quat); xmat[b] = C.q2mat(xquat[b])
    xipos = np.array([xpos[b] + xmat[b] @ A["body_ipos"][b] for b in range(nbody)])
    ximat = np.array([xmat[b] @ C.q2mat(A["body_iquat"][b]) for b in range(nbody)])
    gb = A["geom_bodyid"]
    geom_xpos = np.array([xpos[gb[g]] + xmat[gb[g]] @ A["geom_pos"][g] for g in range(len(gb))]).reshape(-1, 3)
    geom_xmat = np.array([xmat[gb[g]] @ C.q2mat(A["geom_quat"][g]) for g in range(len(gb))]).reshape(-1, 3, 3)
    sb = A["site_bodyid"]
    site_xpos = np.array([xpos[sb[s]] + xmat[sb[s]] @ A["site_pos"][s] for s in range(len(sb))]).reshape(-1, 3)
    return dict(xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, ximat=ximat, xanchor=xanchor, xaxis=xaxis,
                geom_xpos=geom_xpos, geom_xmat=geom_xmat, site_xpos=site_xpos)


def jac(m, kin, point, body):
    """Translational and rotational Jacobian (3×nv each) of a world point fixed to `body`."""
    A = m.arrays
    nv = len(A["dof_bodyid"])
    jp, jr = np.zeros((3, nv)), np.zeros((3, nv))
    b = body
    while b > 0:
        for j in range(A["body_jntadr"][b], A["body_jntadr"][b] + A["body_jntnum"][b]) if A["body_jntnum"][b] else []:
            t, da = A["jnt_type"][j], A["jnt_dofadr"][j]
            if t == C.JNT_FREE:
                jp[:, da:da + 3] = np.eye(3)
                for k in range(3):
                    ax = kin["xmat"][b][:, k]
                    jr[:, da + 3 + k] = ax; jp[:, da + 3 + k] = np.cross(ax, point - kin["xpos"][b])
            elif t == C.JNT_BALL:
                for k in range(3):
                    ax = kin["xmat"][b][:, k]
                    jr[:, da + k] = ax; jp[:, da + k] = np.cross(ax, point - kin["xanchor"][j])
            elif t == C.JNT_SLIDE:
                jp[:, da] = kin["xaxis"][j]
            else:
                jr[:, da] = kin["xaxis"][j]; jp[:, da] = np.cross(kin["xaxis"][j], point - kin["xanchor"][j])
        b = A["body_parentid"][b]
    return jp, jr


def _is_intersect(p1, p2, p3, p4):
    det = (p4[1] - p3[1]) * (p2[0] - p1[0]) - (p4[0] - p3[0]) * (p2[1] - p1[1])
    if abs(det) < MINVAL:
        return False
    a = ((p4[0] - p3[0]) * (p1[1] - p3[1]) - (p4[1] - p3[1]) * (p1[0] - p3[0])) / det
    b = ((p2[0] - p1[0]) * (p1[1] - p3[1]) - (p2[1] - p1[1]) * (p1[0] - p3[0])) / det
    return 0 <= a <= 1 and 0 <= b <= 1


def wrap_circle(d, sd, rad):
    """2-D tangent wrap of the segment d[0:2]→d[2:4] around a circle at the origin.
    Returns (arc length, 4 tangent-point coords) or (-1, None) when the path is straight."""
    d = np.asarray(d, dtype=float)
    sq0, sq1, sqr = d[0] ** 2 + d[1] ** 2, d[2] ** 2 + d[3] ** 2, rad * rad
    dif = d[2:] - d[:2]
    dd = dif @ dif
    if sq0 < sqr or sq1 < sqr or rad < MINVAL or dd < MINVAL:
        return -1.0, None
    a = min(1.0, max(0.0, -(dif @ d[:2]) / dd))
    near = a * dif + d[:2]
    if near @ near > sqr and (sd is None or sd @ near >= 0):
        return -1.0, None
    sols, good = [], []
    for sgn in (1.0, -1.0):
        r0, r1 = np.sqrt(sq0 - sqr), np.sqrt(sq1 - sqr)
        sol = np.array([(d[0] * sqr + sgn * rad * d[1] * r0) / sq0, (d[1] * sqr - sgn * rad * d[0] * r0) / sq0,
                        (d[2] * sqr - sgn * rad * d[3] * r1) / sq1, (d[3] * sqr + sgn * rad * d[2] * r1) / sq1])
        if sd is not None:
            mid = sol[:2] + sol[2:]
            mid = mid / max(np.linalg.norm(mid), MINVAL)
            g = mid @ sd
        else:
            t = sol[:2] - sol[2:]
            g = -(t @ t)
        if _is_intersect(d[:2], sol[:2], d[2:], sol[2:]):
            g = -10000.0
        sols.append(sol); good.append(g)
    sol = sols[0] if good[0] > good[1] else sols[1]
    if _is_intersect(d[:2], sol[:2], d[2:], sol[2:]):
        return -1.0, None
    c = np.clip((sol[:2] @ sol[2:]) / sqr, -1, 1)
    return rad * np.arccos(c), sol


def wrap(x0, x1, gpos, gmat, radius, wtype, side):
    """3-D wrap of the path x0→x1 around a sphere / z-axis cylinder.  Returns (wlen, w0, w1) or (-1,…)."""
    p0, p1 = gmat.T @ (x0 - gpos), gmat.T @ (x1 - gpos)
    if np.linalg.norm(p0) < MINVAL or np.linalg.norm(p1) < MINVAL:
        return -1.0, None, None
    if wtype == C.WRAP_SPHERE:
        ax0 = p0 / np.linalg.norm(p0)
        nrm = np.cross(p0, p1)
        n = np.linalg.norm(nrm)
        if n < MINVAL:
            nrm = np.cross(ax0, [1.0, 0, 0] if abs(ax0[0]) < 0.9 else [0, 1.0, 0]); n = np.linalg.norm(nrm)
        nrm = nrm / n
        ax1 = np.cross(nrm, ax0); ax1 /= np.linalg.norm(ax1)
    else:
        ax0, ax1 = np.array([1.0, 0, 0]), np.array([0, 1.0, 0])
    d = np.array([p0 @ ax0, p0 @ ax1, p1 @ ax0, p1 @ ax1])
    sd = None
    if side is not None:
        s = gmat.T @ (side - gpos)
        sd = np.array([s @ ax0, s @ ax1])
        if np.linalg.norm(sd) < radius:
            raise NotImplementedError("inside wrap (sidesite inside the wrapping geom)")
        sd = sd / np.linalg.norm(sd) * radius
    wlen, sol = wrap_circle(d, sd, radius)
    if wlen < 0:
        return -1.0, None, None
    r0 = ax0 * sol[0] + ax1 * sol[1]
    r1 = ax0 * sol[2] + ax1 * sol[3]
    if wtype == C.WRAP_CYLINDER:
        L0 = np.hypot(d[0] - sol[0], d[1] - sol[1]); L1 = np.hypot(d[2] - sol[2], d[3] - sol[3])
        r0[2] = p0[2] + (p1[2] - p0[2]) * L0 / (L0 + wlen + L1)
        r1[2] = p0[2] + (p1[2] - p0[2]) * (L0 + wlen) / (L0 + wlen + L1)
        wlen = np.hypot(wlen, r1[2] - r0[2])
    return wlen, gmat @ r0 + gpos, gmat @ r1 + gpos


def tendon(m, kin, qpos):
    """Tendon lengths and Jacobians (ntendon×nv)."""
    A = m.arrays
    nt, nv = len(A["tendon_adr"]), len(A["dof_bodyid"])
    L, Jt = np.zeros(nt), np.zeros((nt, nv))
    for t in range(nt):
        adr, num = A["tendon_adr"][t], A["tendon_num"][t]
        if A["wrap_type"][adr] == C.WRAP_JOINT:
            for w in range(adr, adr + num):
                j = A["wrap_objid"][w]
                L[t] += A["wrap_prm"][w] * qpos[A["jnt_qposadr"][j]]
                Jt[t, A["jnt_dofadr"][j]] = A["wrap_prm"][w]
            continue
        w = adr
        divisor = 1.0
        while w < adr + num - 1:
            if A["wrap_type"][w] == C.WRAP_PULLEY:
                divisor = A["wrap_prm"][w]; w += 1
                continue
            if A["wrap_type"][w + 1] == C.WRAP_PULLEY:
                w += 1
                continue
            s0 = A["wrap_objid"][w]
            x0, b0 = kin["site_xpos"][s0], A["site_bodyid"][s0]
            segs = []
            if A["wrap_type"][w + 1] in (C.WRAP_SPHERE, C.WRAP_CYLINDER):
                g = A["wrap_objid"][w + 1]; s1 = A["wrap_objid"][w + 2]
                x1, b1 = kin["site_xpos"][s1], A["site_bodyid"][s1]
                sid = int(A["wrap_prm"][w + 1])
                wl, w0, w1 = wrap(x0, x1, kin["geom_xpos"][g], kin["geom_xmat"][g], A["geom_size"][g][0],
                                  A["wrap_type"][w + 1], kin["site_xpos"][sid] if sid >= 0 else None)
                if wl < 0:
                    segs.append((x0, b0, x1, b1))
                else:
                    gbody = A["geom_bodyid"][g]
                    segs.append((x0, b0, w0, gbody)); segs.append((w1, gbody, x1, b1))
                    L[t] += wl / divisor
                w += 2
            else:
                s1 = A["wrap_objid"][w + 1]
                segs.append((x0, b0, kin["site_xpos"][s1], A["site_bodyid"][s1]))
                w += 1
            for (pa, ba, pb, bb) in segs:
                dvec = pb - pa
                dist = np.linalg.norm(dvec)
                L[t] += dist / divisor
                if ba != bb and dist > MINVAL:
                    ja, _ = jac(m, kin, pa, ba); jb, _ = jac(m, kin, pb, bb)
                    Jt[t] += (dvec / dist) @ (jb - ja) / divisor
    return L, Jt


def inertia_matrix(m, kin):
    """Dense joint-space inertia M(q) from world-frame body Jacobians (+ armature)."""
    A = m.arrays
    nv = len(A["dof_bodyid"])
    M = np.diag(A["dof_armature"].astype(float))
    for b in range(1, len(A["body_mass"])):
        if A["body_mass"][b] <= 0 and not A["body_inertia"][b].any():
            continue
        jp, jr = jac(m, kin, kin["xipos"][b], b)
        Iw = kin["ximat"][b] @ np.diag(A["body_inertia"][b]) @ kin["ximat"][b].T
        M += A["body_mass"][b] * jp.T @ jp + jr.T @ Iw @ jr
    return M


def set_constants(m):
    A = m.arrays
    nbody, nv = len(A["body_parentid"]), len(A["dof_bodyid"])
    sub = A["body_mass"].astype(float).copy()
    for b in range(nbody - 1, 0, -1):
        sub[A["body_parentid"][b]] += sub[b]
    A["body_subtreemass"] = sub
    kin = kinematics(m, A["qpos0"])
    M = inertia_matrix(m, kin)
    Minv = np.linalg.inv(M) if nv else np.zeros((0, 0))
    A["stat_meaninertia"] = np.array([max(MINVAL, np.mean(np.diag(M))) if nv else 1.0])
    biw = np.zeros((nbody, 2))
    for b in range(1, nbody):
        if A["body_weldid"][b] == 0:
            continue
        jp, jr = jac(m, kin, kin["xipos"][b], b)
        biw[b, 0] = max(MINVAL, np.trace(jp @ Minv @ jp.T) / 3)
        biw[b, 1] = max(MINVAL, np.trace(jr @ Minv @ jr.T) / 3)
    A["body_invweight0"] = biw
    diw = np.diag(Minv).copy()
    for j in range(len(A["jnt_type"])):
        da, t = A["jnt_dofadr"][j], A["jnt_type"][j]
        if t == C.JNT_BALL:
            diw[da:da + 3] = diw[da:da + 3].mean()
        elif t == C.JNT_FREE:
            diw[da:da + 3] = diw[da:da + 3].mean(); diw[da + 3:da + 6] = diw[da + 3:da + 6].mean()
    A["dof_invweight0"] = diw
    L, Jt = tendon(m, kin, A["qpos0"])
    A["tendon_length0"] = L
    A["tendon_invweight0"] = np.array([max(MINVAL, Jt[t] @ Minv @ Jt[t]) for t in range(len(L))])
    ls = A["tendon_lengthspring"].copy()
    ls[ls < 0] = L[ls < 0]
    A["tendon_lengthspring"] = ls
    acc0 = []
    for i in range(len(A["actuator_trntype"])):
        mom = np.zeros(nv)
        if A["actuator_trntype"][i] == C.TRN_JOINT:
            mom[A["jnt_dofadr"][A["actuator_trnid"][i]]] = A["actuator_gear"][i]
        else:
            mom = Jt[A["actuator_trnid"][i]] * A["actuator_gear"][i]
        acc0.append(np.linalg.norm(Minv @ mom))
    A["actuator_acc0"] = np.array(acc0)
    # sidesite-inside check is static (site and wrapping geom share a body in the supported models)
    return m
