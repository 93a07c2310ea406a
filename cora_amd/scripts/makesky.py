"""``cora-makesky`` driver for the components this package accelerates.

Mirrors the command-line surface of cora/scripts/makesky.py: the frequency-channelisation options
(``FreqState``, :44-198), the map options (``--nside --pol --filename``, :170-198), the ``21cm``
(:313-345), ``gaussianfg`` (:348-390) and ``singlesource`` (:393-409) commands and the map container
written by ``write_map`` (:412-450).  The ``foreground`` / ``galaxy`` / ``pointsource`` commands need the
constrained-galaxy and point-source catalogue models (non-Gaussian, data-file driven), which are not part
of the hot path: they exist here so that scripts fail with a clear message rather than "no such command".

    python -m cora_amd.scripts.makesky 21cm --nside 256 --freq 400 800 64 --freq-mode edge --pol none --filename m.h5
"""
import ast

import click
import numpy as np

_CHIME_BAND = (800.0, 400.0, 1025)


class FreqState:
    """Frequency specification of a run: band (start, stop, number), how start/stop are meant
    (``freq_mode``), optional rebinning and channel selection."""

    def __init__(self):
        self.freq = _CHIME_BAND
        self.channel_range = None
        self.channel_list = None
        self.channel_bin = 1
        self.freq_mode = "centre"

    def _grid(self):
        start, stop, num = self.freq
        if self.freq_mode == "centre":            # first channel centred on `start`, Nyquist channel dropped
            width = abs(stop - start) / num
            nu = np.linspace(start, stop, num, endpoint=False)
        elif self.freq_mode == "centre_nyquist":  # ... Nyquist channel kept
            width = abs((stop - start) / (num - 1))
            nu = np.linspace(start, stop, num, endpoint=True)
        else:                                     # "edge": start/stop are the band edges
            width = (stop - start) / num
            nu = start + width * (np.arange(num) + 0.5)
        if self.channel_bin > 1:                  # rebin before selecting
            nu = nu.reshape(-1, self.channel_bin).mean(axis=1)
            width = width * self.channel_bin
        if self.channel_list is not None:         # the list wins over the range
            nu = nu[self.channel_list]
        elif self.channel_range is not None and self.channel_range[0] is not None:
            nu = nu[self.channel_range[0]:self.channel_range[1]]
        return nu, width

    @property
    def frequencies(self):
        """Channel centres in MHz."""
        return self._grid()[0]

    @property
    def freq_width(self):
        """Channel width in MHz."""
        return self._grid()[1]


class _IntList(click.ParamType):
    """A Python literal list of ints on the command line, e.g. ``--channel-list "[0, 5, 7]"``."""

    name = "frequency list"

    def convert(self, value, param, ctx):
        if isinstance(value, list):
            return value
        try:
            v = ast.literal_eval(value)
        except (SyntaxError, ValueError):
            self.fail('Could not parse "%s" into list.' % value)
        if not isinstance(v, list):
            self.fail('Could not parse "%s" into list.' % value)
        if not all(isinstance(x, int) for x in v):
            self.fail('Not all values were of type "%s"' % repr(int))
        return v


def _store(ctx, param, value):
    setattr(ctx.ensure_object(FreqState), param.name, value)
    return value


_FREQ_OPTIONS = [
    (("--freq",), dict(type=(float, float, int), default=(800.0, 400.0, 1024), metavar="FSTART FSTOP FNUM",
                       help="Start and stop frequency (MHz) and effective number of channels. "
                            "Default is for CHIME: FSTART=800.0, FSTOP=400.0, FNUM=1025")),
    (("--channel-range",), dict(type=(int, int), default=(None, None), metavar="CSTART CSTOP",
                                help="Select a range of frequency channels. Overriden by channel list.")),
    (("--channel-list",), dict(type=_IntList(), default=None, metavar="CHANNEL LIST",
                               help="Select a list of frequency channels. Takes priority over channel range.")),
    (("--channel-bin",), dict(type=int, default=1, metavar="BIN",
                              help="If set, average over BIN channels. The binning is done before channel selection.")),
    (("--freq-mode",), dict(type=click.Choice(["centre", "centre_nyquist", "edge"]), default="centre",
                            help='FSTART/FSTOP are band edges ("edge") or the centres of the first and last channel, '
                                 'with the last (Nyquist) frequency skipped ("centre", default) or included '
                                 '("centre_nyquist").')),
]

_MAP_OPTIONS = [
    (("--nside",), dict(default=256, metavar="NSIDE", help="Set the map resolution (default: 256)")),
    (("--pol",), dict(type=click.Choice(["full", "zero", "none"]), default="full",
                      help="Pick polarisation mode. Full output, zero polarisation, or only return Stokes I (default: full).")),
    (("--filename",), dict(default="map.h5", metavar="FILENAME", help="Output file [default=map.h5]")),
]


def map_options(f):
    """Decorator: frequency + map options; the command receives ``fstate`` first."""
    f = click.make_pass_decorator(FreqState, ensure=True)(f)
    for names, kw in _FREQ_OPTIONS:
        f = click.option(*names, expose_value=False, callback=_store, **kw)(f)
    for names, kw in _MAP_OPTIONS:
        f = click.option(*names, **kw)(f)
    return f


@click.group()
def cli():
    """Generate a map of the low frequency radio sky (MI355X build: Gaussian components)."""


def _not_in_scope(name):
    raise click.ClickException(
        "'%s' needs cora's constrained-galaxy / point-source catalogue models, which are not part of cora_amd "
        "(Gaussian-sky hot path only); use the reference package for this component." % name)


@cli.command()
@map_options
@click.option("--maxflux", default=1e6, type=float)
def foreground(fstate, nside, pol, filename, maxflux):
    """Full foreground sky (galaxy + point sources): not available in cora_amd."""
    _not_in_scope("foreground")


@cli.command()
@map_options
@click.option("--spectral-index", default="md", type=click.Choice(["md", "gsm", "gd"]))
def galaxy(fstate, nside, pol, filename, spectral_index):
    """Milky way only foreground map: not available in cora_amd."""
    _not_in_scope("galaxy")


@cli.command()
@map_options
@click.option("--maxflux", default=1e6, type=float)
def pointsource(fstate, nside, pol, filename, maxflux):
    """Point source only foreground map: not available in cora_amd."""
    _not_in_scope("pointsource")


@cli.command("21cm")
@map_options
@click.option("--eor", is_flag=True, help="Use parameters more suitable for reionisation epoch.")
@click.option("--oversample", type=int,
              help="Oversample in redshift by 2**oversample_z + 1 to approximate finite width bins.")
@click.option("--seed", type=int, default=None, help="Seed of the numpy Generator (cora_amd extension).")
def _21cm(fstate, nside, pol, filename, eor, oversample, seed):
    """Generate a Gaussian simulation of the unresolved 21cm background."""
    from ..signal import corr21cm

    cr = corr21cm.EoR21cm() if eor else corr21cm.Corr21cm()
    cr.nside = nside
    cr.frequencies = fstate.frequencies
    cr.oversample = oversample if oversample is not None else 3
    rng = np.random.default_rng(seed) if seed is not None else None
    sg_map = cr.getpolsky(rng=rng) if pol == "full" else cr.getsky(rng=rng)
    write_map(filename, sg_map, cr.frequencies, fstate.freq_width, pol != "none")


@cli.command()
@map_options
@click.option("--seed", type=int, default=None, help="Seed of the numpy Generator (cora_amd extension).")
def gaussianfg(fstate, nside, pol, filename, seed):
    """Generate a full-sky Gaussian random field for synchrotron emission."""
    from ..core import skysim
    from ..foreground import galaxy as galaxy_mod
    from ..util import hputil

    fsyn = galaxy_mod.FullSkySynchrotron()
    fpol = galaxy_mod.FullSkyPolarisedSynchrotron()
    fsyn.frequencies = fstate.frequencies
    nfreq = len(fsyn.frequencies)
    lmax = 3 * nside
    npol = 4 if pol == "full" else 1
    # block-diagonal covariance over (pol, freq): T from the unpolarised model, E and B from the polarised one, V = 0
    cv_fg = np.zeros((lmax + 1, npol, nfreq, npol, nfreq))
    cv_fg[:, 0, :, 0, :] = skysim.clarray(fsyn.angular_powerspectrum, lmax, fsyn.nu_pixels)
    if pol == "full":
        cv_fg[:, 1, :, 1, :] = skysim.clarray(fpol.angular_powerspectrum, lmax, fsyn.nu_pixels)
        cv_fg[:, 2, :, 2, :] = skysim.clarray(fpol.angular_powerspectrum, lmax, fsyn.nu_pixels)
    cv_fg = cv_fg.reshape(lmax + 1, npol * nfreq, npol * nfreq)
    rng = np.random.default_rng(seed) if seed is not None else None
    alms = skysim.mkfullsky(cv_fg, nside, alms=True, rng=rng).reshape(npol, nfreq, lmax + 1, lmax + 1)
    maps = hputil.sphtrans_inv_sky(alms.transpose((1, 0, 2, 3)), nside)               # [nfreq, npol, npix]
    write_map(filename, maps if pol == "full" else maps[:, 0], fsyn.frequencies, fstate.freq_width, pol != "none")


@cli.command()
@map_options
@click.option("--ra", type=float, default=0, help="RA (in degrees) for source to add.")
@click.option("--dec", type=float, default=0, help="DEC (in degrees) of source to add.")
def singlesource(fstate, nside, pol, filename, ra, dec):
    """Generate a test map with a single source (amplitude I=1) at the given position."""
    from ..util import hputil

    nfreq = len(fstate.frequencies)
    npol = 4 if pol == "full" else 1
    map_ = np.zeros((nfreq, npol, 12 * nside**2), dtype=np.float64)
    map_[:, 0, hputil.ang2pix(nside, ra, dec, lonlat=True)] = 1.0
    write_map(filename, map_, fstate.frequencies, fstate.freq_width, pol != "none")


def map_container(data, freq, fwidth=None, include_pol=True):
    """The datasets/attributes cora's map files hold (scripts/makesky.py:412-450), as a dict:
    ``map`` [freq, pol, pixel] f64, ``index_map/freq`` (centre, width), ``index_map/pol``, ``index_map/pixel``."""
    data = np.asarray(data)
    if data.ndim == 3:
        polmap = ["I", "Q", "U", "V"]
    elif include_pol:
        full = np.zeros((data.shape[0], 4, data.shape[1]), dtype=data.dtype)
        full[:, 0] = data
        data = full
        polmap = ["I", "Q", "U", "V"]
    else:
        data = data[:, np.newaxis, :]
        polmap = ["I"]
    freqmap = np.zeros(len(freq), dtype=[("centre", np.float64), ("width", np.float64)])
    freqmap["centre"][:] = freq
    freqmap["width"][:] = fwidth if fwidth is not None else np.abs(np.diff(freq)[0])
    return {"map": data, "index_map/freq": freqmap, "index_map/pol": np.array(polmap),
            "index_map/pixel": np.arange(data.shape[2])}


def write_map(filename, data, freq, fwidth=None, include_pol=True):
    """Write the map file.  With h5py: the HDF5 layout of the reference (memh5 attributes included).
    Without h5py (it is not a dependency of this package): the same datasets in an ``.npz`` next to the
    requested name (keys with '/' replaced by '__'); the path written is returned."""
    c = map_container(data, freq, fwidth, include_pol)
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is None:
        path = filename + ".npz" if not filename.endswith(".npz") else filename
        np.savez(path, **{k.replace("/", "__"): v for k, v in c.items()},
                 map__axis=np.array(["freq", "pol", "pixel"]))
        return path
    dt = h5py.special_dtype(vlen=str)
    with h5py.File(filename, "w") as f:
        f.attrs["__memh5_distributed_file"] = True
        dset = f.create_dataset("map", data=c["map"])
        dset.attrs["axis"] = np.array(["freq", "pol", "pixel"]).astype(dt)
        dset.attrs["__memh5_distributed_dset"] = True
        for key in ("index_map/freq", "index_map/pol", "index_map/pixel"):
            val = c[key].astype(dt) if key.endswith("pol") else c[key]
            f.create_dataset(key, data=val).attrs["__memh5_distributed_dset"] = False
    return filename


if __name__ == "__main__":
    cli()
