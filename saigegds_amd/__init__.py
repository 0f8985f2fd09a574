"""saigegds_amd -- MI355X-native single-variant association scan of SAIGEgds.

Public surface (mirrors the reference's R API for this path):
    seqAssocGLMM_SPA   host driver of the scan        (R/assoc_single.r:92-334)
    load_modobj        .check_modobj                  (R/saige_main.r:93-111)
    init_nullmod       .init_nullmod                  (R/assoc_single.r:17-67)
The compute lives in libsaigehip.so (include/saigehip.h); there is no CPU path.
"""
from .nullmod import NullModel, ScanModel, init_nullmod, load_modobj  # noqa: F401
from .assoc import GenotypeSource, seqAssocGLMM_SPA  # noqa: F401
from .fitnull import FittedNullModel, glmmHeritability, seqFitNullGLMM_SPA  # noqa: F401
from .aggregate import (AggrParamBeta, pACAT, pACAT2, seqAssocGLMM_spaACAT_O, seqAssocGLMM_spaACAT_V,  # noqa: F401
                        seqAssocGLMM_spaBurden)

__version__ = "0.1.0"
