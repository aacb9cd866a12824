"""Import-path mirror of the reference's modeling/sequential/hstu.py: `from rails_amd.modeling.sequential.hstu import HSTU`."""
from ...hstu import HSTU, TIMESTAMPS_KEY  # noqa: F401
