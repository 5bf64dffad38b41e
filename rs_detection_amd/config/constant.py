"""Dataset class-name tables (/root/reference/python/jdet/config/constant.py:167-223): the DOTA family and the
FAIR1M tables the shipped configs name (configs[3] orcnn_van3 and this fork's s2anet_r101 ms config train on
FAIR1M-1.5's ten coarse classes)."""
DOTA1_CLASSES = ['plane', 'baseball-diamond', 'bridge', 'ground-track-field', 'small-vehicle', 'large-vehicle', 'ship',
                 'tennis-court', 'basketball-court', 'storage-tank', 'soccer-ball-field', 'roundabout', 'harbor',
                 'swimming-pool', 'helicopter']
DOTA1_5_CLASSES = DOTA1_CLASSES + ['container-crane']
DOTA2_CLASSES = DOTA1_5_CLASSES + ['airport', 'helipad']
_AIR = ['Boeing737', 'Boeing747', 'Boeing777', 'Boeing787', 'C919', 'A220', 'A321', 'A330', 'A350', 'ARJ21',
        'other-airplane']
_FAIR_TAIL = ['Passenger Ship', 'Motorboat', 'Fishing Boat', 'Tugboat', 'Engineering Ship', 'Liquid Cargo Ship',
              'Dry Cargo Ship', 'Warship', 'other-ship', 'Small Car', 'Bus', 'Cargo Truck', 'Dump Truck', 'Van',
              'Trailer', 'Tractor', 'Excavator', 'Truck Tractor', 'other-vehicle', 'Basketball Court', 'Tennis Court',
              'Football Field', 'Baseball Field', 'Intersection', 'Roundabout', 'Bridge']
FAIR_CLASSES = _AIR + _FAIR_TAIL
FAIR_CLASSES_ = _AIR + [c.replace(' ', '_') for c in _FAIR_TAIL]      # the file-name-safe spelling (:186-194)
FAIR1M_1_5_CLASSES = ['Airplane', 'Ship', 'Vehicle', 'Basketball_Court', 'Tennis_Court', 'Football_Field',
                      'Baseball_Field', 'Intersection', 'Roundabout', 'Bridge']


def get_classes_by_name(name):
    res = {'DOTA': DOTA1_CLASSES, 'DOTA1': DOTA1_CLASSES, 'DOTA1_5': DOTA1_5_CLASSES, 'DOTA2': DOTA2_CLASSES,
           'FAIR': FAIR_CLASSES_, 'FAIR1M_1_5': FAIR1M_1_5_CLASSES}
    assert name in res, name
    return res[name]
