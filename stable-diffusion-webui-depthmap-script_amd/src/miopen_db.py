"""MIOpen user find-db shipped with the package (no counterpart in the reference; round 6).

The float32 convolutions of LeReS / the pix2pix U-Net (Boost) and the few half-precision library convolutions the other networks
still call are shapes nobody has searched on a fresh machine: MIOpen's default find mode then benchmarks every applicable solver
per shape -- 334 s of wall for BASELINE config 4's first image on an MI355X, 7 s once the results are in the user find-db
(`MIOPEN_FIND_MODE=FAST` avoids the search too, but its immediate-mode picks run the image in 1156 ms instead of 602).  The records
are a 70 KB text file keyed by problem, architecture and MIOpen build: `miopen_db/*.ufdb.txt` holds what the searches of this
package's own networks found on gfx950 with the image's MIOpen; `seed()` copies a file into the user-db directory
(`MIOPEN_USER_DB_PATH`, default `~/.config/miopen`) when no file of that name exists there yet -- another MIOpen build uses another
file name and simply ignores it.  `DS_MIOPEN_SEED=0` switches the seeding off."""
import glob
import os
import shutil

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_done = [False]


def seed():
    if _done[0] or os.environ.get("DS_MIOPEN_SEED", "1") == "0":
        return []
    _done[0] = True
    dst_dir = os.environ.get("MIOPEN_USER_DB_PATH") or os.path.join(os.path.expanduser("~"), ".config", "miopen")
    copied = []
    try:
        for f in sorted(glob.glob(os.path.join(HERE, "miopen_db", "*.ufdb.txt"))):
            dst = os.path.join(dst_dir, os.path.basename(f))
            if not os.path.exists(dst):
                os.makedirs(dst_dir, exist_ok=True)
                tmp = "%s.seed.%d" % (dst, os.getpid())      # several ranks may seed at once: the file appears whole or not at all
                shutil.copyfile(f, tmp)
                os.replace(tmp, dst)
                copied.append(dst)
    except OSError:            # a read-only home: MIOpen searches as before
        pass
    return copied
