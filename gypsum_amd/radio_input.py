"""Recording descriptors (`gypsum/radio_input.py:15-126`): what a file on disk is, so that a provider can be built
from it.  Same class / constructor names as upstream; the vendored recordings themselves do not ship with the
reference, so `INPUT_SOURCES` starts empty and `register_input_source` adds to it."""
from __future__ import annotations

import datetime
from dataclasses import dataclass
from enum import Enum, auto
from pathlib import Path
from typing import List, Optional, Type

import numpy as np

PRN_CHIP_COUNT = 1023               # constants.py:7
PRN_REPETITIONS_PER_SECOND = 1000   # constants.py:10


class InputFileType(Enum):
    Raw = auto()
    Wav = auto()
    GnuRadioRecording = auto()


@dataclass
class InputFileInfo:
    path: Path
    format: InputFileType
    sdr_sample_rate: int
    utc_start_time: datetime.datetime
    sample_component_data_type: Type[np.number]

    @classmethod
    def gnu_radio_recording(cls, path: Path, sample_rate: int, utc_start_time: datetime.datetime) -> "InputFileInfo":
        # GNU Radio recordings are a pair of interleaved 32-bit floats per IQ sample (radio_input.py:37-44)
        return cls(path=Path(path), format=InputFileType.GnuRadioRecording, sdr_sample_rate=sample_rate,
                   sample_component_data_type=np.float32, utc_start_time=utc_start_time)

    @classmethod
    def _nx(cls, path: Path, multiple: int) -> "InputFileInfo":
        return cls.gnu_radio_recording(path, sample_rate=PRN_CHIP_COUNT * PRN_REPETITIONS_PER_SECOND * multiple,
                                       utc_start_time=datetime.datetime.utcfromtimestamp(0))

    @classmethod
    def gnu_radio_recording_2x(cls, path: Path, utc_start_time: Optional[datetime.datetime] = None) -> "InputFileInfo":
        """2.046 MHz, radio_input.py:46-62 (the utc_start_time argument is ignored upstream too)."""
        return cls._nx(path, 2)

    @classmethod
    def gnu_radio_recording_8x(cls, path: Path, utc_start_time: Optional[datetime.datetime] = None) -> "InputFileInfo":
        """8.184 MHz, radio_input.py:64-78."""
        return cls._nx(path, 8)

    @classmethod
    def gnu_radio_recording_16x(cls, path: Path, utc_start_time: Optional[datetime.datetime] = None) -> "InputFileInfo":
        """16.368 MHz, radio_input.py:80-94."""
        return cls._nx(path, 16)

    @classmethod
    def raw_int8(cls, path: Path, sample_rate: int, utc_start_time: Optional[datetime.datetime] = None) -> "InputFileInfo":
        """Interleaved int8 I/Q as HackRF / RTL-SDR tools write it (not upstream: the reference only reads float32;
        its `np.fromfile(dtype=...)` path gives the raw integer values, which is what the native ingest reproduces)."""
        return cls(path=Path(path), format=InputFileType.Raw, sdr_sample_rate=sample_rate,
                   utc_start_time=utc_start_time or datetime.datetime.utcfromtimestamp(0), sample_component_data_type=np.int8)


INPUT_SOURCES: List[InputFileInfo] = []


def register_input_source(info: InputFileInfo) -> InputFileInfo:
    INPUT_SOURCES.append(info)
    return info


def get_input_source_by_file_name(name: str) -> InputFileInfo:
    """radio_input.py:111-126, same errors."""
    matching = [x for x in INPUT_SOURCES if x.path.name == name]
    if len(matching) == 0:
        raise FileNotFoundError(f'No input file named "{name}" found. '
                                f"Make sure you describe this file and its signal structure in radio_input.py::INPUT_SOURCES.")
    if len(matching) > 1:
        raise RuntimeError(f'Found more than one file named "{name}".')
    return matching[0]
