"""contrastiveseg_amd -- MI355X-native contrastive-segmentation training hot path.

Importing the package points MIOpen at the tuned solver records shipped in `miopen_db/` (found once on an MI355X
with MIOpen's find mode for the convolutions that dominate the HRNet-W48 step): with them MIOpen's
immediate mode picks the measured-fastest solver (e.g. the xdlops implicit-GEMM weight-gradient kernel for the
3x3 720->720 head convolution instead of the 2.7x slower Winograd default) without paying a 20+ minute find on
every fresh machine. The records are copied to a per-process scratch directory because MIOpen opens its user
database read-write."""
import atexit
import os
import shutil
import tempfile

_PKG = os.path.dirname(os.path.abspath(__file__))


def configure_miopen(force=False):
    if "MIOPEN_USER_DB_PATH" in os.environ and not force:
        return os.environ["MIOPEN_USER_DB_PATH"]
    src = os.path.join(_PKG, "miopen_db")
    if not os.path.isdir(src) or os.environ.get("CSEG_NO_MIOPEN_DB"):
        return None
    dst = os.path.join(tempfile.gettempdir(), "cseg_miopen_db_%d_%s_%d" % (
        os.getuid(), os.environ.get("LOCAL_RANK", "0"), os.getpid()))
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(src):
        if f.endswith(".txt"):
            shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    atexit.register(shutil.rmtree, dst, True)
    return dst


configure_miopen()
