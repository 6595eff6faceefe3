"""`stringdecomposer.main` of the reference (bin/stringdecomposer imports it): alias of the MI355X build's driver."""
from stringdecomposer_amd.main import *  # noqa: F401,F403
from stringdecomposer_amd.main import main, run, convert_tsv, convert_read, load_fasta  # noqa: F401
