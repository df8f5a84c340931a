"""
Quick-look figures used by the FWI drivers: velocity model with acquisition markers, velocity
difference, shot record and generic image.  Same function names, arguments and axes conventions
(x in km to the right, depth or time downwards) as the reference's seismic/plotting.py:15-165;
matplotlib is imported on first use so that the package works without it.
"""
import numpy as np

__all__ = ['plot_perturbation', 'plot_velocity', 'plot_shotrecord', 'plot_image']


def _pyplot():
    try:
        import matplotlib.pyplot as plt
    except ImportError as exc:                      # pragma: no cover
        raise ImportError("the plotting helpers need matplotlib") from exc
    return plt


def _draw(image, extent=None, label=None, colorbar=True, show=True, **imshow_kw):
    """imshow + optional colour bar of the same height on the right."""
    plt = _pyplot()
    handle = plt.imshow(image, extent=extent, **imshow_kw)
    axes = plt.gca()
    if colorbar:
        from mpl_toolkits.axes_grid1 import make_axes_locatable
        cax = make_axes_locatable(axes).append_axes("right", size="5%", pad=0.05)
        bar = plt.colorbar(handle, cax=cax)
        if label:
            bar.set_label(label)
        plt.sca(axes)
    if show:
        plt.show()
    return handle


def _section_extent(model):
    """[x0, x1, z1, z0] in km of the physical domain (depth axis pointing down)."""
    size = 1.e-3 * np.array(model.domain_size)
    return [model.origin[0], model.origin[0] + size[0], model.origin[1] + size[1], model.origin[1]]


def _physical(model, data):
    data = np.asarray(data)
    return data[model.physical_slices] if data.shape == model.grid.shape else data


def plot_perturbation(model, model1, colorbar=True, show=True):
    """Velocity difference model - model1 of two 2-D models (plotting.py:15-48)."""
    plt = _pyplot()
    dv = np.transpose(np.asarray(model.vp.data)) - np.transpose(np.asarray(model1.vp.data))
    plt.xlabel('X position (km)')
    plt.ylabel('Depth (km)')
    return _draw(dv, _section_extent(model), 'Velocity perturbation (km/s)', colorbar, show,
                 cmap='jet', vmin=dv.min(), vmax=dv.max())


def plot_velocity(model, source=None, receiver=None, colorbar=True, cmap="jet", show=True):
    """2-D velocity section with optional source (red dots) and receiver (green diamonds) markers,
    coordinates in metres (plotting.py:51-104)."""
    plt = _pyplot()
    extent = _section_extent(model)
    field = _physical(model, model.vp.data)
    handle = _draw(np.transpose(field), extent, 'Velocity (km/s)', colorbar, False, cmap=cmap,
                   vmin=np.min(field), vmax=np.max(field))
    if receiver is not None:
        plt.scatter(1e-3 * np.asarray(receiver)[:, 0], 1e-3 * np.asarray(receiver)[:, 1], s=25, c='green', marker='D')
    if source is not None:
        plt.scatter(1e-3 * np.asarray(source)[:, 0], 1e-3 * np.asarray(source)[:, 1], s=25, c='red', marker='o')
    plt.xlabel('X position (km)')
    plt.ylabel('Depth (km)')
    plt.xlim(extent[0], extent[1])
    plt.ylim(extent[2], extent[3])
    if show:
        plt.show()
    return handle


def plot_shotrecord(rec, model, t0, tn, colorbar=True, show=True):
    """Shot record (time, receiver) in grey scale clipped at a tenth of its maximum; t0, tn in ms
    (plotting.py:107-137)."""
    plt = _pyplot()
    rec = np.asarray(rec)
    clip = np.max(rec) / 10.
    extent = [model.origin[0], model.origin[0] + 1e-3 * model.domain_size[0], 1e-3 * tn, t0]
    plt.xlabel('X position (km)')
    plt.ylabel('Time (s)')
    return _draw(rec, extent, None, colorbar, show, cmap='gray', vmin=-clip, vmax=clip)


def plot_image(data, vmin=None, vmax=None, colorbar=True, cmap="gray", show=True):
    """Any (x, z) image such as a gradient or an inverted model (plotting.py:140-165)."""
    data = np.asarray(data)
    return _draw(np.transpose(data), None, None, colorbar, show, cmap=cmap,
                 vmin=vmin or 0.9 * np.min(data), vmax=vmax or 1.1 * np.max(data))
