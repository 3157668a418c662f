from .raven import RavenAdamW
from .titan import TitanAdamW

__all__ = ["RavenAdamW", "TitanAdamW"]
