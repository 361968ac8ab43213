from .SO_FJSSP import SO_FJSSP_Environment, BatchedSOFJSSP  # noqa: F401
