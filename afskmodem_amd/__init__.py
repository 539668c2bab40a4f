"""afskmodem_amd -- lavajuno/afskmodem's Receiver/Transmitter API with the receiver
hot path running as hand-written HIP kernels on MI355X (gfx950).

    import afskmodem_amd as afskmodem
    afskmodem.Transmitter(1200).save("Hello World!", "x.wav")
    afskmodem.Receiver(1200).load("x.wav")          # demodulated on the GPU

Batched / device-resident entry points live in ``afskmodem_amd.batch``; multi-GPU
sharding in ``afskmodem_amd.dist``; synthetic workloads in ``afskmodem_amd.synth``.
"""
# Log level (0: Debug, 1: Info, 2: Warn, 3: Error, 4: Fatal) -- same global as the reference (:14)
LOG_LEVEL = 0

from .modem import ECC, Log, Receiver, SoundInput, SoundOutput, Transmitter, Waveforms, load_batch  # noqa: E402

__all__ = ["ECC", "Log", "Receiver", "SoundInput", "SoundOutput", "Transmitter", "Waveforms",
           "LOG_LEVEL", "load_batch"]
