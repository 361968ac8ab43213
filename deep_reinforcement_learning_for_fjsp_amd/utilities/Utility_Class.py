"""The one class of the reference's utilities/Utility_Class.py the environment
path needs (the plotting / Pareto helpers there are out of scope)."""


class MyError(Exception):
    """utilities/Utility_Class.py:272-276: exception carrying `.message`."""

    def __init__(self, message):
        self.message = message
        super().__init__(self.message)
