"""Empty stand-in (utilities/Utility_Class.py:8: `from openpyxl.styles import numbers`)."""


class _Anything(object):
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return _Anything()


numbers = _Anything()
