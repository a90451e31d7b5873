"""
Batched device solve on PyTorch-ROCm tensors: the host side of ``okx_solve_batch``.

``DeviceProgram`` uploads one constraint program to the current GPU and exposes
``solve`` (B independent sweep-step problems, or geometry-major chains), ``eval``
(residual + dense Jacobian, for parity tests) and ``rebind`` (per-geometry design targets
for perturbed-hardpoint ensembles).  Tensors stay in HBM; nothing here copies to the host.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._abi import (EVAL_COLUMNS, EVAL_MAX_PIVOT, EVAL_MIN_PIVOT, EVAL_RATE_RACK_Y, EVAL_RATE_WHEEL_CENTER_X,
                   EVAL_AXLE_METRICS, EVAL_AXLE_ROLES, EVAL_TANGENT_FLAGS, INFO_CONVERGED, INFO_DTYPE, INFO_FAILED, INFO_RESIDUAL_EXCEEDED, TANGENT_INFO_DTYPE,
                   HostProgram, SolveOpts)
from .program import ConstraintProgram


@dataclass
class BatchResult:
    positions: torch.Tensor | None  # [B, n_out, 3] float64, device (``output="records"``, the default)
    info_raw: torch.Tensor  # [B, 40] uint8, device (okx_info records)
    free: torch.Tensor | None = None  # [B, n_free, 3] float64, device (``output="free"``): the solved free points alone

    def info(self) -> np.ndarray:
        """Host copy of the per-problem records as a structured array (``_abi.INFO_DTYPE``)."""
        return self.info_raw.cpu().numpy().view(INFO_DTYPE).reshape(-1)

    def host(self) -> tuple:
        """
        ``(positions [B, n_out, 3] or None, info records)`` on the host after ONE wait: both copies are queued behind the
        launch (torch's current stream, where the solve was put) into pinned buffers - the host allocator caches the blocks,
        so a repeated sweep allocates nothing - and the stream is synchronised once, instead of ``.cpu()`` twice (each a
        synchronous copy through a staging buffer).  The arrays own their memory for as long as they live.
        """
        if self.info_raw.device.type != "cuda":
            return (None if self.positions is None else self.positions.numpy()), self.info()
        staged = []
        for tensor in (self.positions, self.info_raw):
            if tensor is None:
                staged.append(None)
                continue
            pinned = torch.empty(tensor.shape, dtype=tensor.dtype, pin_memory=True)
            pinned.copy_(tensor, non_blocking=True)
            staged.append(pinned)
        torch.cuda.current_stream(self.info_raw.device).synchronize()
        positions = None if staged[0] is None else staged[0].numpy()
        return positions, staged[1].numpy().view(INFO_DTYPE).reshape(-1)

    @staticmethod
    def converged(info: np.ndarray) -> np.ndarray:
        return (info["flags"] & INFO_CONVERGED) != 0

    @staticmethod
    def accepted(info: np.ndarray) -> np.ndarray:
        """Converged and within the residual tolerance (reference ``solver.py:726-747``)."""
        f = info["flags"]
        return ((f & INFO_CONVERGED) != 0) & ((f & (INFO_RESIDUAL_EXCEEDED | INFO_FAILED)) == 0)


N_METRICS = 19  # OKX_METRIC_COUNT


@dataclass
class EvaluatedResult(BatchResult):
    """
    What ``DeviceProgram.solve_evaluated`` / ``evaluate`` return: the solve's own results plus the epilogue's
    (``okx_solve_evaluated_batch``): ``eval [B, 1 + T, 24]`` - row 0 the metric values and the tangent solve's pivots,
    row 1 + t the derivatives along target t and the driver rates - and, when asked for, ``tangents [B, T, n_out, 3]``.
    The properties are views, nothing is copied.
    """

    eval: torch.Tensor | None = None
    tangents: torch.Tensor | None = None

    def corner(self, side_index: int) -> "EvaluatedResult":
        """Of a composed axle's result (``eval [B, 1 + T, 64]``, ``okx.h`` OKX_EVAL_AXLE_*): one corner's block as a
        corner-shaped result (a view; 0 = left, 1 = right; the tangent solve's pivots sit in the left block)."""
        lo = EVAL_COLUMNS * side_index
        return EvaluatedResult(self.positions, self.info_raw, self.free, self.eval[:, :, lo:lo + EVAL_COLUMNS], self.tangents)

    @property
    def axle_metrics(self) -> torch.Tensor:
        """``[B, 7]`` axle-scope metrics in ``metrics.AXLE_METRIC_NAMES`` order (a composed axle's result)."""
        return self.eval[:, 0, EVAL_AXLE_METRICS:EVAL_AXLE_METRICS + 7]

    @property
    def role_values(self) -> torch.Tensor:
        """``[B, 8]`` rotation / hardware role values; ``role_rates`` ``[B, T, 8]`` their rates along every target's tangent."""
        return self.eval[:, 0, EVAL_AXLE_ROLES:EVAL_AXLE_ROLES + 8]

    @property
    def role_rates(self) -> torch.Tensor:
        return self.eval[:, 1:, EVAL_AXLE_ROLES:EVAL_AXLE_ROLES + 8]

    @property
    def metrics(self) -> torch.Tensor:
        """``[B, 19]`` metric values in ``metrics.METRIC_NAMES`` order (NaN where the reference reports None)."""
        return self.eval[:, 0, :N_METRICS]

    @property
    def derivatives(self) -> torch.Tensor:
        """``[B, T, 19]``: d metric / d target."""
        return self.eval[:, 1:, :N_METRICS]

    @property
    def wheel_center_rates(self) -> torch.Tensor:
        """``[B, T, 3]``: d wheel centre / d target (the ``hub_z`` driver and the ``wheel_center_x`` response)."""
        return self.eval[:, 1:, EVAL_RATE_WHEEL_CENTER_X:EVAL_RATE_WHEEL_CENTER_X + 3]

    @property
    def rack_rates(self) -> torch.Tensor:
        """``[B, T]``: d rack pickup y / d target (NaN without a rack)."""
        return self.eval[:, 1:, EVAL_RATE_RACK_Y]

    def tangent_info(self) -> np.ndarray:
        """The tangent solves' health as ``_abi.TANGENT_INFO_DTYPE`` records (host)."""
        row = self.eval[:, 0, EVAL_MIN_PIVOT:EVAL_TANGENT_FLAGS + 1].cpu().numpy()
        out = np.zeros(row.shape[0], dtype=TANGENT_INFO_DTYPE)
        out["min_pivot"], out["max_pivot"], out["flags"] = row[:, 0], row[:, 1], row[:, 2].astype(np.int32)
        return out


def _ptr(t: torch.Tensor | None) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _as_f64(t, device) -> torch.Tensor:
    """A contiguous float64 tensor ON the device: always a snapshot of host data (synchronous copy), whatever its kind."""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(np.asarray(t, dtype=np.float64))
    return t.to(device=device, dtype=torch.float64).contiguous()


def _is_mapped_host(t) -> bool:
    """A contiguous pinned host tensor: device-accessible memory, which ``okx.h`` accepts for its ``d_*`` pointers."""
    return isinstance(t, torch.Tensor) and not t.is_cuda and t.is_contiguous() and t.is_pinned()


def _check_buffer(name: str, t, device, zero_copy: bool) -> None:
    """An output buffer of a launch must live where the kernel can write it: on the launch device, or - opted into with
    ``zero_copy=True`` - in pinned host memory.  Pageable host memory would fault the GPU (XNACK is off)."""
    if t is None or (t.is_cuda and t.device == device):
        return
    if zero_copy and _is_mapped_host(t):
        return
    raise ValueError(f"{name} must be a tensor on {device}" + (" or a contiguous pinned host tensor" if zero_copy else
                     " (pinned host buffers are accepted with zero_copy=True)") + f", got {t.device}"
                     + ("" if t.is_cuda or not t.is_pinned() else ", pinned"))


class DeviceProgram:
    """A constraint program resident on one GPU."""

    def __init__(self, program: ConstraintProgram, device: torch.device | str | None = None, wait_for_kernels: bool = True):
        if not torch.cuda.is_available():
            raise RuntimeError(
                "open_kinematics_amd needs a ROCm GPU: torch.cuda.is_available() is False "
                "(there is no CPU fallback for the solve path)"
            )
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.program = program
        self.host = HostProgram(program)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.okx_program_create(self.host.byref(), C.byref(handle)), "okx_program_create")
        self._handle = handle
        # Generated kernels that are not in the kernel cache are compiled on a host thread (okx.h: tiered start).  The batch
        # API waits for them by default - its callers time and compare kernels; the drop-in passes False and starts
        # solving at once on the interpreter kernels.
        if wait_for_kernels:
            self.wait_ready()
        self._predictor: bool | None = None  # None: no fit attempted yet (fit_predictor)
        self.predictor_box = None
        self._predictor_note = ""

    @property
    def kernel(self) -> str:
        """``"quad"`` when the runtime-specialised kernel is loaded, else ``"wave"`` (generic interpreter)."""
        return self.lib.okx_program_kernel(self._handle).decode()

    @property
    def kernel_note(self) -> str:
        """Why the quad kernel is not in use (empty when it is)."""
        return self.lib.okx_program_kernel_note(self._handle).decode()

    @property
    def shares_first_step(self) -> bool:
        """Chain heads of the own geometry take their first step from the shared first-step table (their ``nfev`` omits it)."""
        return bool(self.lib.okx_program_shares_first_step(self._handle))

    @property
    def ready(self) -> bool:
        """False while the program's generated kernels are still being compiled (the interpreter kernels serve it)."""
        with torch.cuda.device(self.device):
            return bool(self.lib.okx_program_ready(self._handle, 0))

    def wait_ready(self) -> None:
        """Block until the compile job (if any) has finished and the program has switched to its generated kernels."""
        with torch.cuda.device(self.device):
            _lib.check(0 if self.lib.okx_program_ready(self._handle, 1) >= 0 else -1, "okx_program_ready")

    @property
    def has_cold_body(self) -> bool:
        """Independent solves on the own geometry run the quad kernel's cold body (``okx_quad_cold_u``)."""
        return bool(self.lib.okx_program_has_cold_body(self._handle))

    @property
    def lane_threshold(self) -> int:
        """Batch size from which auto selection uses the lane kernel (one lane per problem); -1: the program has none."""
        return int(self.lib.okx_program_lane_threshold(self._handle))

    @property
    def lane_bodies(self) -> int:
        """Bodies of the lane kernel auto selection uses: bit0 independent solves, bit1 chains (0: none)."""
        return int(self.lib.okx_program_lane_bodies(self._handle))

    @property
    def lane_note(self) -> str:
        """Why the program has no lane kernel (empty when it has one)."""
        return self.lib.okx_program_lane_note(self._handle).decode()

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self.lib.okx_program_destroy(self._handle)
            self._handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def default_opts(self) -> SolveOpts:
        opts = SolveOpts()
        self.lib.okx_default_opts(C.byref(opts))
        return opts

    def _prepare(
        self,
        targets,
        *,
        geom_pos: torch.Tensor | None = None,
        geom_row_param: torch.Tensor | None = None,
        steps_per_geometry: int = 0,
        chain: bool = False,
        chain_len: int | None = None,
        max_iter: int | None = None,
        step_tol: float | None = None,
        lambda0: float | None = None,
        ftol: float | None = None,
        grad_tol: float | None = None,
        kernel: int | str | None = None,
        residual_tolerance: float | None = None,
        out: torch.Tensor | None = None,
        info_out: torch.Tensor | None = None,
        predictor: bool | str | None = None,
        confirm_full_pass: bool | None = None,
        shared_first_step: bool | None = None,
        output: str = "records",
        zero_copy: bool = False,
    ) -> BatchResult:
        """
        Solve ``B`` problems; ``targets`` is ``[B, T]`` of absolute target scalars.

        ``predictor``: chain heads start from the polynomial model of ``okx_program_fit_predictor`` instead of
        the design state (own-geometry launches of the quad kernel).  ``None`` (default) = use a model that
        ``fit_predictor`` has fitted, never fit one implicitly; ``True`` = require it (fitted over this launch's target
        box if there is none yet); ``False`` = cold starts; ``"all"`` = every chain step starts from the model (instead
        of the secant extrapolation).

        ``shared_first_step`` (default on): chain heads take their first LM step from the per-geometry table of the
        design state instead of running that (batch-invariant) pass themselves (``okx_solve_opts``).

        ``confirm_full_pass``: always end a solve on a computed correction ``<= step_tol`` (``okx_solve_opts``).

        ``output`` (``okx_solve_opts.output``): ``"records"`` (default) writes every output point ``[B, n_out, 3]``;
        ``"free"`` the solved free points alone ``[B, n_free, 3]`` in ``free_point`` order (``BatchResult.free``;
        ``expand`` rebuilds the records, bit-identical) - what a PCIe link or an all-gather wants to carry; ``"none"``
        only the info records.  ``out`` is the buffer of whichever is written.

        ``zero_copy=True``: ``targets``, ``out`` and ``info_out`` given as contiguous PINNED host tensors are handed to the
        kernel as they are (``okx.h``: ``d_*`` pointers may be device-accessible host memory) - it reads the targets from
        and stores its results into the caller's buffers over PCIe, no copy commands.  The caller then owns the
        synchronisation: the buffers must stay untouched until the launch has completed on its stream.  Without the
        flag host inputs are snapshotted by a synchronous copy and host output buffers are refused.

        ``chain_len`` groups consecutive problems into warm-started chains walked by one
        wavefront each (``1`` independent cold starts, ``-1`` one chain per resident wavefront,
        ``None`` follows ``chain``: whole-sweep chain or independent).
        """
        p = self.program
        if zero_copy and _is_mapped_host(targets) and targets.dtype == torch.float64:
            targets = targets.reshape(-1, max(p.n_targets, 1))  # read in place, over PCIe
        else:
            targets = _as_f64(targets, self.device).reshape(-1, max(p.n_targets, 1))
        _check_buffer("out", out, self.device, zero_copy)
        _check_buffer("info_out", info_out, self.device, zero_copy)
        b = targets.shape[0] if p.n_targets > 0 else int(targets.numel())
        opts = self.default_opts()
        opts.chain = 1 if chain else 0
        opts.steps_per_geometry = int(steps_per_geometry)
        if chain_len is not None:
            opts.chain_len = int(chain_len)
        if max_iter is not None:
            opts.max_iter = int(max_iter)
        if step_tol is not None:
            opts.step_tol = float(step_tol)
        if lambda0 is not None:
            opts.lambda0 = float(lambda0)
        if ftol is not None:
            opts.ftol = float(ftol)
        if grad_tol is not None:   # > 0: max |J^T r|; < 0: MINPACK's gtol, max_j |(J^T r)_j| / (|J_j| |r|) <= -grad_tol (okx.h)
            opts.grad_tol = float(grad_tol)
        if kernel is not None:
            opts.kernel = {"auto": 0, "single": 1, "packed": 2, "quad": 3, "lane": 4}.get(kernel, kernel)
        if residual_tolerance is not None:
            opts.residual_tolerance = float(residual_tolerance)
        if confirm_full_pass is not None:
            opts.confirm_full_pass = 1 if confirm_full_pass else 0
        if shared_first_step is not None:
            opts.shared_first_step = 1 if shared_first_step else 0
        if predictor is not False and geom_pos is None and opts.kernel in (0, 3):
            # The model is never fitted behind the caller's back (the fit is a synchronous ~8 ms step: node solves, D2H,
            # host fit): predictor=None uses a model that fit_predictor() has put there, True / "all" fit one over this
            # launch's target box if there is none yet (later launches clamp to that box; refit with fit_predictor).
            fit_now = self._predictor is None and predictor is not None
            if self.fit_predictor(targets if fit_now else None, required=bool(predictor)):
                opts.predictor = 2 if predictor == "all" else 1
        if geom_pos is not None:
            geom_pos = _as_f64(geom_pos, self.device)
            geom_row_param = _as_f64(geom_row_param, self.device)
            g = geom_pos.shape[0]
            if geom_pos.shape[1:] != (p.n_points, 3) or geom_row_param.shape != (g, p.n_rows, 8):
                raise ValueError("geometry table has the wrong shape")
            if steps_per_geometry <= 0 or g * steps_per_geometry != b:
                raise ValueError("B must equal n_geometries * steps_per_geometry")
        mode = {"records": 0, "free": 1, "none": 2}[output]
        opts.output = mode
        if mode == 2:
            out = None
        elif out is None:
            out = torch.empty((b, p.n_out if mode == 0 else p.n_free, 3), dtype=torch.float64, device=self.device)
        elif out.shape != (b, p.n_out if mode == 0 else p.n_free, 3) or out.dtype != torch.float64 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float64 [B, n_out, 3] ([B, n_free, 3] with output='free') tensor")
        if info_out is None:
            info_out = torch.empty((b, INFO_DTYPE.itemsize), dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        args = (self._handle, C.byref(opts), b, _ptr(targets), _ptr(geom_pos), _ptr(geom_row_param),
                _ptr(out), _ptr(info_out), C.c_void_p(stream))
        keep = (opts, targets, geom_pos, geom_row_param)  # tensors / structs the raw pointers refer to
        return args, keep, BatchResult(out if mode == 0 else None, info_out, out if mode == 1 else None)

    # ---- the evaluated solve: tangents and metrics as the solve kernels' epilogue (okx.h) ----

    def enable_evaluation(self, roles) -> None:
        """
        ``okx_program_enable_evaluation``: generate / load the program's evaluated kernels for the metric roles
        ``roles`` (``metrics.corner_roles`` / ``make_roles``).  The role points are compiled in (a first call with new
        role points compiles for 10 - 60 s unless ``__graft_entry__.build()`` has filled the cache); the numbers of the
        roles are kernel arguments, a call that only changes them costs nothing.
        """
        from .metrics import AxleRoles

        with torch.cuda.device(self.device):
            if isinstance(roles, AxleRoles):  # a composed axle: both corners' roles + rotation / hardware roles (okx_axle_roles)
                _lib.check(self.lib.okx_program_enable_axle_evaluation(self._handle, C.byref(roles)), "okx_program_enable_axle_evaluation")
            else:
                _lib.check(self.lib.okx_program_enable_evaluation(self._handle, C.byref(roles)), "okx_program_enable_evaluation")
        self._roles = roles

    @property
    def evaluation(self) -> int:
        """bit0: evaluated kernels are loaded, bit1: with a lane form (one lane per problem) for large batches."""
        return int(self.lib.okx_program_evaluation(self._handle))

    @property
    def evaluation_note(self) -> str:
        return self.lib.okx_program_evaluation_note(self._handle).decode()

    @property
    def eval_columns(self) -> int:
        """Columns per row of ``eval`` for the evaluation last enabled: 24 (a corner), 64 (a composed axle), 0 (none)."""
        return int(self.lib.okx_program_eval_columns(self._handle))

    def _eval_buffers(self, b: int, tangents, eval_out):
        p = self.program
        columns = self.eval_columns or EVAL_COLUMNS
        if eval_out is None:
            eval_out = torch.empty((b, 1 + p.n_targets, columns), dtype=torch.float64, device=self.device)
        elif eval_out.shape != (b, 1 + p.n_targets, columns) or eval_out.dtype != torch.float64 or not eval_out.is_contiguous():
            raise ValueError(f"eval_out must be a contiguous float64 [B, 1 + T, {columns}] tensor")
        tan = None
        if isinstance(tangents, torch.Tensor):
            tan = tangents
            if tan.shape != (b, p.n_targets, p.n_out, 3) or tan.dtype != torch.float64 or not tan.is_contiguous():
                raise ValueError("tangents must be a contiguous float64 [B, T, n_out, 3] tensor")
        elif tangents:
            tan = torch.empty((b, p.n_targets, p.n_out, 3), dtype=torch.float64, device=self.device)
        _check_buffer("eval_out", eval_out, self.device, False)
        _check_buffer("tangents", tan, self.device, False)
        return tan, eval_out

    def _prepare_evaluated(self, targets, roles, tangents, eval_out, **kw):
        if roles is not None:
            self.enable_evaluation(roles)
        if not self.evaluation:
            raise RuntimeError("solve_evaluated needs metric roles: pass roles= or call enable_evaluation first")
        args, keep, res = self._prepare(targets, **kw)
        b = res.info_raw.shape[0]
        tan, ev = self._eval_buffers(b, tangents, eval_out)
        args = args[:-1] + (_ptr(tan), _ptr(ev), args[-1])
        return args, keep, EvaluatedResult(res.positions, res.info_raw, res.free, ev, tan)

    def solve_evaluated(self, targets, *, roles=None, tangents=False, eval_out=None, **kw) -> EvaluatedResult:
        """
        ``okx_solve_evaluated_batch``: ``solve`` (same keywords) whose kernels end every problem with the evaluation
        epilogue - the reference's ``solve_evaluated_sweep`` (``core/sweep.py:248-270``) for a batch, ONE launch:
        solution-manifold tangents at the converged state, the corner metric catalog and its derivative along every
        target's tangent.  ``output="none"`` returns metrics and derivative columns without ever writing positions.
        ``tangents=True`` (or a buffer) also materialises ``[B, T, n_out, 3]``.
        """
        args, _keep, result = self._prepare_evaluated(targets, roles, tangents, eval_out, **kw)
        with torch.cuda.device(self.device):
            rc = self.lib.okx_solve_evaluated_batch(*args)
        _lib.check(rc, "okx_solve_evaluated_batch")
        return result

    def plan_evaluated(self, targets, *, roles=None, tangents=False, eval_out=None, **kw):
        """``plan`` for ``solve_evaluated``: a zero-argument callable whose only work is the C-ABI call."""
        args, keep, result = self._prepare_evaluated(targets, roles, tangents, eval_out, **kw)
        fn, check = self.lib.okx_solve_evaluated_batch, _lib.check

        def launch() -> EvaluatedResult:
            rc = fn(*args)
            if rc != 0:
                check(rc, "okx_solve_evaluated_batch")
            return result

        launch.keep = keep
        return launch

    def evaluate(self, positions, *, roles=None, tangents=False, eval_out=None, geom_pos=None, geom_row_param=None,
                 steps_per_geometry: int = 0, info_raw=None) -> EvaluatedResult:
        """
        ``okx_evaluate_batch``: the same epilogue on GIVEN solved states ``positions [B, n_out, 3]`` - the reference's
        ``evaluate_solved_sweep`` (``core/sweep.py:217-245``) in one launch instead of tangents -> metrics.
        """
        if roles is not None:
            self.enable_evaluation(roles)
        if not self.evaluation:
            raise RuntimeError("evaluate needs metric roles: pass roles= or call enable_evaluation first")
        p = self.program
        pos = _as_f64(positions, self.device).reshape(-1, p.n_out, 3)
        b = pos.shape[0]
        if geom_pos is not None:
            geom_pos = _as_f64(geom_pos, self.device)
            geom_row_param = _as_f64(geom_row_param, self.device)
            if geom_pos.shape[1:] != (p.n_points, 3) or geom_row_param.shape != (geom_pos.shape[0], p.n_rows, 8):
                raise ValueError("geometry table has the wrong shape")
            if steps_per_geometry <= 0 or geom_pos.shape[0] * steps_per_geometry != b:
                raise ValueError("B must equal n_geometries * steps_per_geometry")
        tan, ev = self._eval_buffers(b, tangents, eval_out)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            rc = self.lib.okx_evaluate_batch(self._handle, b, int(steps_per_geometry), _ptr(pos), _ptr(geom_pos),
                                             _ptr(geom_row_param), _ptr(tan), _ptr(ev), C.c_void_p(stream))
        _lib.check(rc, "okx_evaluate_batch")
        # (``info_raw``: the info records of the solve these states came from, carried through to the result)
        info = info_raw if info_raw is not None else torch.zeros((0, INFO_DTYPE.itemsize), dtype=torch.uint8, device=self.device)
        return EvaluatedResult(pos, info, None, ev, tan)

    def solve(self, targets, **kw) -> BatchResult:
        """See ``_prepare`` for the arguments: validates, then launches ``okx_solve_batch`` once."""
        args, _keep, result = self._prepare(targets, **kw)
        with torch.cuda.device(self.device):
            rc = self.lib.okx_solve_batch(*args)
        _lib.check(rc, "okx_solve_batch")
        return result

    def plan_launch(self, n_problems: int, *, steps_per_geometry: int = 0, geometry_tables: bool = False, evaluated: bool = False,
                    chain: bool = False, chain_len: int | None = None, kernel: int | str | None = None,
                    predictor: bool | str | None = False, **_ignored) -> tuple:
        """
        ``okx_plan_launch``: ``(kernel, chain_len)`` a launch of ``n_problems`` with these keywords would resolve to -
        the values to pass as ``kernel=`` / ``chain_len=`` to every piece of a batch that is cut into several launches
        and must keep the bits of the single launch (auto selection depends on the problem count).
        """
        opts = self.default_opts()
        opts.chain = 1 if chain else 0
        opts.steps_per_geometry = int(steps_per_geometry)
        if chain_len is not None:
            opts.chain_len = int(chain_len)
        if kernel is not None:
            opts.kernel = {"auto": 0, "single": 1, "packed": 2, "quad": 3, "lane": 4}.get(kernel, kernel)
        if predictor not in (False, None) or (predictor is None and self._predictor):
            opts.predictor = 1
        out2 = (C.c_int32 * 2)()
        with torch.cuda.device(self.device):
            rc = self.lib.okx_plan_launch(self._handle, C.byref(opts), int(n_problems), 1 if geometry_tables else 0,
                                          1 if evaluated else 0, C.byref(out2))
        _lib.check(rc, "okx_plan_launch")
        return {1: "single", 2: "packed", 3: "quad", 4: "lane"}[out2[0]], int(out2[1])

    def plan(self, targets, **kw):
        """
        Pre-bound launch for a hot loop: validates and converts the arguments once (same keywords
        as ``solve``) and returns a zero-argument callable whose only work is the
        ``okx_solve_batch`` call on the stream that was current when the plan was made — a few
        microseconds of host time per launch, so back-to-back launches of a ~40 us kernel stay
        GPU-bound.  Pass ``out=`` / ``info_out=`` to fix the output buffers.
        """
        args, keep, result = self._prepare(targets, **kw)
        fn, check = self.lib.okx_solve_batch, _lib.check

        def launch() -> BatchResult:
            # the launch goes to the plan's stream, which carries its device: no device switch here
            rc = fn(*args)
            if rc != 0:
                check(rc, "okx_solve_batch")
            return result

        launch.keep = keep
        return launch

    def eval(self, x, targets, jac: bool = True):
        """Residuals ``[B, m]`` and dense Jacobians ``[B, m, n]`` at free vectors ``x [B, n]``."""
        p = self.program
        x = _as_f64(x, self.device).reshape(-1, p.n_vars)
        b = x.shape[0]
        targets = _as_f64(targets, self.device).reshape(-1, max(p.n_targets, 1))
        if targets.shape[0] == 1 and b > 1:
            targets = targets.expand(b, -1).contiguous()
        r = torch.empty((b, p.n_residuals), dtype=torch.float64, device=self.device)
        j = torch.empty((b, p.n_residuals, p.n_vars), dtype=torch.float64, device=self.device) if jac else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            rc = self.lib.okx_eval_batch(self._handle, b, _ptr(x), _ptr(targets), _ptr(r), _ptr(j), C.c_void_p(stream))
        _lib.check(rc, "okx_eval_batch")
        return r, j

    def normal_equations(self, x, targets):
        """Test hook: ``J^T J [B, n, n]`` and ``J^T r [B, n]`` exactly as the solver forms them."""
        p = self.program
        x = _as_f64(x, self.device).reshape(-1, p.n_vars)
        b = x.shape[0]
        targets = _as_f64(targets, self.device).reshape(-1, max(p.n_targets, 1))
        if targets.shape[0] == 1 and b > 1:
            targets = targets.expand(b, -1).contiguous()
        r = torch.empty((b, p.n_residuals), dtype=torch.float64, device=self.device)
        ata = torch.empty((b, p.n_vars, p.n_vars), dtype=torch.float64, device=self.device)
        atr = torch.empty((b, p.n_vars), dtype=torch.float64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            rc = self.lib.okx_debug_normal_equations(
                self._handle, b, _ptr(x), _ptr(targets), _ptr(r), _ptr(ata), _ptr(atr), C.c_void_p(stream)
            )
        _lib.check(rc, "okx_debug_normal_equations")
        return r, ata, atr

    def tangents(self, positions, *, geom_pos=None, geom_row_param=None, steps_per_geometry: int = 0):
        """
        Solution-manifold tangents of ``B`` solved states (reference ``compute_state_tangents``,
        ``sensitivity.py:57-143``): ``positions [B, n_out, 3]`` as returned by ``solve`` ->
        ``(tangents [B, T, n_out, 3], info)`` with ``tangents[b, t, k] = d point_k / d target_t`` and
        ``info`` a structured array (``_abi.TANGENT_INFO_DTYPE``: pivot range of the LDL^T of
        ``J^T J``, ok / rank-deficient flags).  Stays in HBM; one kernel launch.
        """
        p = self.program
        pos = _as_f64(positions, self.device).reshape(-1, p.n_out, 3)
        b = pos.shape[0]
        if geom_pos is not None:
            geom_pos = _as_f64(geom_pos, self.device)
            geom_row_param = _as_f64(geom_row_param, self.device)
            if geom_pos.shape[1:] != (p.n_points, 3) or geom_row_param.shape != (geom_pos.shape[0], p.n_rows, 8):
                raise ValueError("geometry table has the wrong shape")
            if steps_per_geometry <= 0 or geom_pos.shape[0] * steps_per_geometry != b:
                raise ValueError("B must equal n_geometries * steps_per_geometry")
        tan = torch.empty((b, p.n_targets, p.n_out, 3), dtype=torch.float64, device=self.device)
        tinfo = torch.empty((b, TANGENT_INFO_DTYPE.itemsize), dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            rc = self.lib.okx_tangent_batch(self._handle, b, int(steps_per_geometry), _ptr(pos), _ptr(geom_pos),
                                            _ptr(geom_row_param), _ptr(tan), _ptr(tinfo), C.c_void_p(stream))
        _lib.check(rc, "okx_tangent_batch")
        return tan, tinfo

    @staticmethod
    def tangent_info(tinfo: torch.Tensor) -> np.ndarray:
        return tinfo.cpu().numpy().view(TANGENT_INFO_DTYPE).reshape(-1)

    def quad_eval(self, x, targets, lam: float = 0.0, lane: bool = False):
        """
        Test hook for the runtime-specialised kernels: residuals ``[B, m]``, ``J^T J [B, n, n]``
        (symmetrised from the lane-owned lower rows), ``J^T r [B, n]`` and the damped step
        ``dx = -(J^T J + lam I)^-1 J^T r`` from its in-register LDL^T, at free vectors ``x [B, n]``.
        ``lane=True``: the same from the lane kernel's code (one lane per problem).
        """
        p = self.program
        x = _as_f64(x, self.device).reshape(-1, p.n_vars)
        b = x.shape[0]
        targets = _as_f64(targets, self.device).reshape(-1, max(p.n_targets, 1))
        if targets.shape[0] == 1 and b > 1:
            targets = targets.expand(b, -1).contiguous()
        n = p.n_vars
        r = torch.empty((b, p.n_residuals), dtype=torch.float64, device=self.device)
        ata = torch.zeros((b, n, n), dtype=torch.float64, device=self.device)
        atr = torch.empty((b, n), dtype=torch.float64, device=self.device)
        dx = torch.empty((b, n), dtype=torch.float64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            rc = (self.lib.okx_debug_lane_eval if lane else self.lib.okx_debug_quad_eval)(
                self._handle, b, _ptr(x), _ptr(targets), float(lam), _ptr(r), _ptr(ata), _ptr(atr), _ptr(dx),
                C.c_void_p(stream),
            )
        _lib.check(rc, "okx_debug_quad_eval")
        # the kernel writes each structurally non-zero off-diagonal block once (rows of the block that
        # is eliminated later) and the diagonal blocks in full: mirror what was not written
        ata = torch.where(ata != 0.0, ata, ata.transpose(1, 2))
        return r, ata, atr, dx

    def fit_predictor(self, targets=None, *, lo=None, hi=None, degree: int = 0, required: bool = False) -> bool:
        """
        Fit the chain-head predictor (``okx_program_fit_predictor``) over the box of ``targets [B, T]`` (their
        per-target min / max) or an explicit ``lo`` / ``hi``; without arguments an existing fit is kept.
        Returns whether the program has one; programs without a quad kernel, with free points outside the
        output list or with a node that does not converge simply go without unless ``required``.
        """
        if targets is not None or lo is not None:
            if self.kernel != "quad" or self.program.n_targets == 0:
                self._predictor, self._predictor_note = False, "no quad kernel / no targets"
            else:
                if lo is None:
                    t = _as_f64(targets, self.device).reshape(-1, self.program.n_targets)
                    lo, hi = t.min(dim=0).values.cpu().numpy(), t.max(dim=0).values.cpu().numpy()
                lo = np.ascontiguousarray(lo, dtype=np.float64).reshape(self.program.n_targets)
                hi = np.ascontiguousarray(hi, dtype=np.float64).reshape(self.program.n_targets)
                stream = torch.cuda.current_stream(self.device).cuda_stream
                with torch.cuda.device(self.device):
                    rc = self.lib.okx_program_fit_predictor(self._handle, lo.ctypes.data_as(C.c_void_p),
                                                            hi.ctypes.data_as(C.c_void_p), int(degree), C.c_void_p(stream))
                self._predictor = rc == 0
                self._predictor_note = "" if rc == 0 else _lib.last_error()
                self.predictor_box = (lo, hi) if rc == 0 else None
        if required and not self._predictor:
            raise RuntimeError(f"no chain-head predictor for this program: {self._predictor_note or 'not fitted'}")
        return bool(self._predictor)

    @property
    def free_out_index(self) -> torch.Tensor:
        """Output-list index of every free point (device int64 ``[n_free]``); raises when one is not an output."""
        if getattr(self, "_free_out", None) is None:
            out = [int(k) for k in self.program.out_point]
            try:
                idx = [out.index(int(p)) for p in self.program.free_point]
            except ValueError:
                raise ValueError("a free point of this program is not among its output points") from None
            self._free_out = torch.as_tensor(idx, dtype=torch.int64, device=self.device)
        return self._free_out

    def expand(self, free: torch.Tensor, out: torch.Tensor | None = None, geom_pos: torch.Tensor | None = None,
               steps_per_geometry: int = 0) -> torch.Tensor:
        """
        ``okx_expand_positions_batch``: free coordinates ``[B, n_free, 3]`` (``free_point`` order, e.g.
        ``positions[:, dp.free_out_index]``) -> output positions ``[B, n_out, 3]``: fixed points from the design
        state (or ``geom_pos``), derived points re-evaluated.
        """
        p = self.program
        free = _as_f64(free, self.device).reshape(-1, p.n_free, 3)
        b = free.shape[0]
        if out is None:
            out = torch.empty((b, p.n_out, 3), dtype=torch.float64, device=self.device)
        elif out.shape != (b, p.n_out, 3) or out.dtype != torch.float64 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float64 [B, n_out, 3] tensor")
        if geom_pos is not None:
            geom_pos = _as_f64(geom_pos, self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            rc = self.lib.okx_expand_positions_batch(self._handle, b, int(steps_per_geometry), _ptr(free), _ptr(geom_pos),
                                                     _ptr(out), C.c_void_p(stream))
        _lib.check(rc, "okx_expand_positions_batch")
        return out

    def ensemble_targets(self, geom_pos: torch.Tensor, relative) -> torch.Tensor:
        """
        Absolute targets ``[G * S, T]`` of an ensemble from per-step RELATIVE displacements ``[S, T]``: every
        geometry's own design coordinate along each target direction plus the displacement — the reference's
        relative target mode applied per geometry (``convert_targets_to_absolute``, ``solver.py:584-627``).
        ``geom_pos [G, P, 3]`` comes from ``rebind``.  Stays in HBM.
        """
        p = self.program
        rel = _as_f64(relative, self.device).reshape(-1, p.n_targets)
        dirs = torch.as_tensor(np.asarray(p.tgt_dir, dtype=np.float64), device=self.device)         # [T, 3]
        pts = torch.as_tensor(np.asarray(p.tgt_point, dtype=np.int64), device=self.device)          # [T]
        base = (geom_pos[:, pts, :] * dirs[None]).sum(dim=2)                                           # [G, T]
        return (base[:, None, :] + rel[None]).reshape(-1, p.n_targets).contiguous()

    def rebind(self, hardpoints):
        """Per-geometry design positions ``[G, P, 3]`` and row parameters ``[G, Mc, 8]``."""
        p = self.program
        hp = _as_f64(hardpoints, self.device).reshape(-1, p.n_points, 3)
        g = hp.shape[0]
        pos = torch.empty_like(hp)
        rq = torch.empty((g, p.n_rows, 8), dtype=torch.float64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            rc = self.lib.okx_rebind_design(self._handle, g, _ptr(hp), _ptr(pos), _ptr(rq), C.c_void_p(stream))
        _lib.check(rc, "okx_rebind_design")
        return pos, rq
