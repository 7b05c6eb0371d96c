"""Structure learning, mirror of sorobn/structure.py (`chow_liu`); see learning.py."""
from .learning import chow_liu, mutual_information

__all__ = ["chow_liu", "mutual_information"]
