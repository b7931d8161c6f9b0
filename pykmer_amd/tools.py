"""Drop-in module name: `from pykmer_amd.tools import Header, Timer` works like the reference's
`from tools import Header, Timer` (indexer.py:13, merger.py:33).  The implementation is header.py."""
from .header import Header, HeaderVars, Timer, gen_checksum, stats_from_hist256  # noqa: F401
