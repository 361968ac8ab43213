from .SO_FJSSP import SO_FJSSP_Environment, BatchedSOFJSSP  # noqa: F401
from .MO_FJSSP_discretes import MO_FJSSP_Environment, BatchedMOFJSSP  # noqa: F401
from .SO_SFJSP import SO_SFJSP_Environment, BatchedSOSFJSP  # noqa: F401
from .MO_DFJSP_breakdown import MO_DFJSP_Environment, BatchedMODFJSP  # noqa: F401
from .SO_DFJSP import SO_DFJSP_Environment, BatchedSODFJSP  # noqa: F401
