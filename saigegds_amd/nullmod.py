"""Null-model object handling for the single-variant scan.

Mirrors the R side of the reference for this path:
  * ``load_modobj``   <- ``.check_modobj``  (R/saige_main.r:93-111)
  * ``init_nullmod``  <- ``.init_nullmod``  (R/assoc_single.r:17-67)

The flattened model (``ScanModel``) holds exactly the arrays that
``saige_score_test_init`` pins in C globals (src/saige_main.cpp:103-150).
All K x N matrices are kept "sample-major": element (k, i) lives at
``a[i, k]`` of a C-contiguous [N, K] numpy array, which is byte-identical to
the column-major K x N matrices the reference passes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Optional, Sequence

import numpy as np

from .rds import read_rdata, RList, read_rds


class ModelError(ValueError):
    pass


@dataclass
class NullModel:
    """Fields of ``ClassSAIGE_NullModel`` used by the scan
    (man/seqFitNullGLMM_SPA.Rd:75-93)."""
    trait_type: str            # "binary" | "quantitative"
    tau: np.ndarray            # [2]
    fitted_values: np.ndarray  # [N] GLMM mu
    sample_id: list            # [N]
    var_ratio: np.ndarray      # ratios of the random markers
    y: np.ndarray              # obj.noK$y
    V: np.ndarray              # obj.noK$V
    X1: np.ndarray             # obj.noK$X1       [N, K]
    XV: np.ndarray             # obj.noK$XV       [K, N]
    XXVX_inv: np.ndarray       # obj.noK$XXVX_inv [N, K]
    coefficients: Optional[np.ndarray] = None
    variant_id: Optional[np.ndarray] = None


def _as_f64(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def modobj_from_rlist(m: RList) -> NullModel:
    if "ClassSAIGE_NullModel" not in m.rclass():
        raise ModelError("'modobj' should be an object of class ClassSAIGE_NullModel.")
    nk = m["obj.noK"]
    tt = m["trait.type"]
    tt = tt[0] if isinstance(tt, (list, tuple)) else str(tt)
    vr = m["var.ratio"]
    ratio = vr["ratio"] if isinstance(vr, RList) else vr
    sid = m["sample.id"]
    sid = list(sid) if isinstance(sid, list) else np.asarray(sid).tolist()
    return NullModel(
        trait_type=tt, tau=_as_f64(m["tau"]), fitted_values=_as_f64(m["fitted.values"]),
        sample_id=sid, var_ratio=_as_f64(ratio), y=_as_f64(nk["y"]), V=_as_f64(nk["V"]),
        X1=_as_f64(nk["X1"]), XV=_as_f64(nk["XV"]), XXVX_inv=_as_f64(nk["XXVX_inv"]),
        coefficients=_as_f64(m["coefficients"]) if "coefficients" in m else None,
        variant_id=np.asarray(m["variant.id"]) if "variant.id" in m else None)


def load_modobj(modobj: Any, verbose: bool = False) -> NullModel:
    """``.check_modobj``: accept a model object or an ``.rds`` file name."""
    if isinstance(modobj, NullModel):
        return modobj
    if isinstance(modobj, RList):
        return modobj_from_rlist(modobj)
    if isinstance(modobj, str):
        if verbose:
            print(f"    load the null model from '{modobj}'")
        low = modobj.lower()
        if low.endswith(".rds"):
            return modobj_from_rlist(read_rds(modobj))
        if low.endswith(".rda") or low.endswith(".rdata"):
            # modobj <- get(load(modobj)), R/saige_main.r:98-100: the first object of the file
            objs = read_rdata(modobj)
            if not objs:
                raise ModelError(f"'{modobj}' holds no object.")
            first = next(iter(objs.values()))
            if not isinstance(first, RList):
                raise ModelError(f"the object in '{modobj}' is not a list.")
            return modobj_from_rlist(first)
        raise ModelError("It should be an RData, RDS or gds file.")
    raise ModelError("'modobj' should be a NullModel, a parsed RDS list or a file name.")


@dataclass
class ScanModel:
    """The list ``.init_nullmod`` builds (R/assoc_single.r:28-48), minus the
    scratch buffers, which live on the device."""
    trait_type: str
    n: int
    k: int
    tau: np.ndarray
    y: np.ndarray
    mu: np.ndarray
    y_mu: np.ndarray
    mu2: np.ndarray
    t_XXVX_inv: np.ndarray    # [N, K]
    XV: np.ndarray            # [N, K]
    t_XVX_inv_XV: np.ndarray  # [N, K]
    t_X: np.ndarray           # [N, K]
    XVX: np.ndarray           # [K, K]
    S_a: np.ndarray           # [K]
    var_ratio: float
    maf: float
    mac: float
    missing: float
    spa_pval: float

    @property
    def quant(self) -> bool:
        return self.trait_type == "quantitative"


def init_nullmod(modobj: NullModel, ii: Sequence[int], maf: float, mac: float,
                 missing: float, spa_pval: float, var_ratio: float) -> ScanModel:
    """``.init_nullmod(modobj, ii, maf, mac, missing, spa.pval, var.ratio)``.

    ``ii`` are 0-based positions into the model's sample order, one per GDS
    sample (``match(sid, modobj$sample.id)`` in the reference, minus one).
    """
    if not math.isfinite(var_ratio):
        raise ModelError("Invalid variance ratio in the SAIGE model.")
    ii = np.asarray(ii, dtype=np.int64)
    y = modobj.y
    mu = modobj.fitted_values
    X1 = modobj.X1[ii, :]
    y_i, mu_i = y[ii], mu[ii]
    y_mu = (y - mu)[ii]
    mu2 = (mu * (1 - mu))[ii]
    xxvx = modobj.XXVX_inv[ii, :]
    if modobj.trait_type == "binary":
        XVX = X1.T @ (X1 * mu2[:, None])
    elif modobj.trait_type == "quantitative":
        XVX = X1.T @ X1
    else:
        raise ModelError(f"Invalid 'modobj$trait.type': {modobj.trait_type}.")
    S_a = (X1 * y_mu[:, None]).sum(axis=0)
    return ScanModel(
        trait_type=modobj.trait_type, n=int(ii.size), k=int(X1.shape[1]),
        tau=_as_f64(modobj.tau), y=_as_f64(y_i), mu=_as_f64(mu_i), y_mu=_as_f64(y_mu),
        mu2=_as_f64(mu2), t_XXVX_inv=_as_f64(xxvx), XV=_as_f64(modobj.XV[:, ii].T),
        t_XVX_inv_XV=_as_f64(xxvx * modobj.V[ii][:, None]), t_X=_as_f64(X1),
        XVX=_as_f64(XVX), S_a=_as_f64(S_a), var_ratio=float(var_ratio),
        maf=float(maf), mac=float(mac), missing=float(missing), spa_pval=float(spa_pval))
