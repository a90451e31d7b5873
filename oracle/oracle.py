"""
ctypes front end of the CPU oracle (``oracle/okx_oracle.c``).  TEST INFRASTRUCTURE ONLY:
imported by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg,
never by ``open_kinematics_amd`` itself.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

from open_kinematics_amd._abi import HostProgram, ProgramDesc
from open_kinematics_amd.program import ConstraintProgram

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("OKX_ORACLE_LIB") or os.path.join(_HERE, "libokx_oracle.so")  # override: sanitizer build (tests)
_f64p = C.POINTER(C.c_double)


class OracleOpts(C.Structure):
    _fields_ = [
        ("ftol", C.c_double),
        ("xtol", C.c_double),
        ("gtol", C.c_double),
        ("residual_tolerance", C.c_double),
        ("max_nfev", C.c_int32),
        ("warm_start", C.c_int32),
    ]


ORACLE_INFO_DTYPE = np.dtype(
    [
        ("max_residual", "<f8"),
        ("nfev", "<i4"),
        ("njev", "<i4"),
        ("minpack_info", "<i4"),
        ("success", "<i4"),
    ]
)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (also called from ``__graft_entry__.build``)."""
    src = os.path.join(_HERE, "okx_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "okx.h")
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libokx_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if "OKX_ORACLE_LIB" not in os.environ:
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.okx_oracle_eval.restype = C.c_int32
        _lib.okx_oracle_eval.argtypes = [C.POINTER(ProgramDesc), C.c_int64, _f64p, _f64p, _f64p, _f64p]
        _lib.okx_oracle_positions.restype = C.c_int32
        _lib.okx_oracle_positions.argtypes = [C.POINTER(ProgramDesc), _f64p, _f64p]
        _lib.okx_oracle_sweep.restype = C.c_int32
        _lib.okx_oracle_sweep.argtypes = [
            C.POINTER(ProgramDesc),
            C.POINTER(OracleOpts),
            C.c_int64,
            _f64p,
            _f64p,
            _f64p,
            _f64p,
            C.c_void_p,
        ]
        _lib.okx_oracle_rebind.restype = C.c_int32
        _lib.okx_oracle_rebind.argtypes = [C.POINTER(ProgramDesc), _f64p, _f64p, _f64p]
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(_f64p)


@dataclass
class SweepResult:
    positions: np.ndarray  # [S, n_out, 3]
    x: np.ndarray  # [S, n]
    info: np.ndarray  # structured ORACLE_INFO_DTYPE [S]
    first_failed_step: int  # -1 if all accepted


class Oracle:
    """CPU oracle bound to one constraint program."""

    def __init__(self, program: ConstraintProgram):
        self.program = program
        self.host = HostProgram(program)

    def eval(self, x: np.ndarray, targets: np.ndarray, jac: bool = True):
        p = self.program
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        b = x.shape[0]
        t = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(targets), (b, p.n_targets)), dtype=np.float64)
        r = np.empty((b, p.n_residuals))
        j = np.empty((b, p.n_residuals, p.n_vars)) if jac else None
        rc = lib().okx_oracle_eval(self.host.byref(), b, _p(x), _p(t), _p(r), _p(j) if jac else None)
        if rc != 0:
            raise RuntimeError(f"okx_oracle_eval failed: {rc}")
        return r, j

    def positions(self, x: np.ndarray) -> np.ndarray:
        p = self.program
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty((p.n_points, 3))
        rc = lib().okx_oracle_positions(self.host.byref(), _p(x), _p(out))
        if rc != 0:
            raise RuntimeError(f"okx_oracle_positions failed: {rc}")
        return out

    def sweep(
        self,
        targets: np.ndarray,
        ftol: float = 1e-5,
        xtol: float = 1e-9,
        gtol: float = 1e-9,
        residual_tolerance: float = 1e-3,
        warm_start: bool = True,
        max_nfev: int = 0,
        x_start: np.ndarray | None = None,
    ) -> SweepResult:
        """``solve_suspension_sweep`` (reference ``solver.py:654-776``) on absolute targets."""
        p = self.program
        t = np.ascontiguousarray(targets, dtype=np.float64).reshape(-1, p.n_targets)
        s = t.shape[0]
        pos = np.empty((s, p.n_out, 3))
        xs = np.empty((s, p.n_vars))
        info = np.zeros(s, dtype=ORACLE_INFO_DTYPE)
        opts = OracleOpts(ftol, xtol, gtol, residual_tolerance, max_nfev, 1 if warm_start else 0)
        x0 = None
        if x_start is not None:
            x0 = np.ascontiguousarray(x_start, dtype=np.float64)
        rc = lib().okx_oracle_sweep(
            self.host.byref(),
            C.byref(opts),
            s,
            _p(t),
            _p(x0) if x0 is not None else None,
            _p(pos),
            _p(xs),
            info.ctypes.data_as(C.c_void_p),
        )
        if rc < 0:
            if rc == -3:
                raise ValueError(
                    f"System is underdetermined (n_vars={p.n_vars} > m_res={p.n_residuals})."
                )
            raise RuntimeError(f"okx_oracle_sweep failed: {rc}")
        return SweepResult(pos, xs, info, rc - 1)

    def tangents(self, x: np.ndarray):
        """
        ``compute_state_tangents`` (reference ``sensitivity.py:57-143``) restated with numpy at the
        free vector ``x``: the analytical Jacobian (C oracle), plus the two smooth pins per
        point-on-line row (``_degenerate_constraint_pins``, ``sensitivity.py:146-174``) when the
        program carries the reference's softnorm line rows, ``numpy.linalg.lstsq`` against one unit
        right-hand side per target row, then the forward-mode velocity of every derived point
        (the reference's dual-number pass, ``points/derived/definitions.py:24-180``).
        Returns ``(vel [T, P, 3] for ALL points, rank, singular_values)``.
        """
        from open_kinematics_amd import program as pm

        p = self.program
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        _, jac = self.eval(x[None], np.zeros((1, p.n_targets)))
        rows = [jac[0]]
        block_of_point = {int(pt): k for k, pt in enumerate(p.free_point)}
        for i in range(p.n_rows):
            if int(p.row_type[i]) != pm.ROW_POINT_ON_LINE or int(p.row_pts[i][0]) not in block_of_point:
                continue
            d = p.row_param[i][3:6] / np.linalg.norm(p.row_param[i][3:6])
            least = np.zeros(3)
            least[int(np.argmin(np.abs(d)))] = 1.0
            n1 = np.cross(d, least)
            n1 /= np.linalg.norm(n1)
            n2 = np.cross(d, n1)
            for normal in (n1, n2):
                row = np.zeros(p.n_vars)
                off = 3 * block_of_point[int(p.row_pts[i][0])]
                row[off:off + 3] = normal
                rows.append(row[None])
        jaug = np.vstack(rows)
        rhs = np.zeros((jaug.shape[0], p.n_targets))
        for t in range(p.n_targets):
            rhs[p.n_rows + t, t] = 1.0
        q, _res, rank, sv = np.linalg.lstsq(jaug, rhs, rcond=None)
        pos = self.positions(x)
        vel = np.zeros((p.n_targets, p.n_points, 3))
        for k, pt in enumerate(p.free_point):
            vel[:, int(pt)] = q[3 * k:3 * k + 3].T
        for e in range(p.n_derived):
            kind, out = int(p.dop_type[e]), int(p.dop_out[e])
            a, b, c3 = (int(v) for v in p.dop_pts[e][:3])
            par = float(p.dop_param[e])
            for t in range(p.n_targets):
                v = vel[t]
                if kind == pm.DOP_MIDPOINT:  # definitions.py:76-89
                    v[out] = v[a] + (v[b] - v[a]) / 2.0
                elif kind == pm.DOP_ALONG:  # definitions.py:24-33: base + normalize(p_b - p_c) * par
                    w, dw = pos[b] - pos[c3], v[b] - v[c3]
                    n = np.linalg.norm(w)
                    u = w / n
                    v[out] = v[a] + par * (dw - u * (u @ dw)) / n
                else:  # contact patch, definitions.py:36-73: wc + normalize(-Z - ((-Z).ax) ax) * R
                    w, dw = pos[c3] - pos[b], v[c3] - v[b]
                    n = np.linalg.norm(w)
                    ax = w / n
                    dax = (dw - ax * (ax @ dw)) / n
                    down = np.array([0.0, 0.0, -1.0])
                    ga, dga = down @ ax, down @ dax
                    wd = down - ga * ax
                    dwd = -dga * ax - ga * dax
                    wn = np.linalg.norm(wd)
                    wu = wd / wn
                    v[out] = v[a] + par * (dwd - wu * (wu @ dwd)) / wn
        return vel, int(rank), sv

    def rebind(self, hardpoints: np.ndarray):
        p = self.program
        hp = np.ascontiguousarray(hardpoints, dtype=np.float64).reshape(p.n_points, 3)
        pos = np.empty((p.n_points, 3))
        rp = np.empty((p.n_rows, 8))
        rc = lib().okx_oracle_rebind(self.host.byref(), _p(hp), _p(pos), _p(rp))
        if rc != 0:
            raise RuntimeError(f"okx_oracle_rebind failed: {rc}")
        return pos, rp
