"""rsim-hip: the MI355X-native backend of robosuite's env.step() hot path.  `make` is the suite.make()-shaped entry (factory.py)."""
from .factory import make  # noqa: F401
