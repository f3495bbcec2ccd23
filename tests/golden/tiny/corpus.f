s89 s44 s1 s2 s78 s0 s57 s50 s7 s2 s2 s1 s6 s9 s12 s155 s49 s18
s1 s0 s79 s7 s11 s2 s38 s0 s4 s0 s0 s132 s22
s25 s31 s19 s136 s3 s4 s1 s0 s1 s99 s64 s0 s16 s7 s15
s17 s0 s23 s19 s58 s52 s3 s32 s75 s87 s0 s0 s21 s1 s13 s116
s5 s0 s66 s11 s4 s11
s7 s0 s2 s2 s32 s6 s0 s155 s85
s1 s7 s81 s41 s60 s41 s30 s68 s25 s35 s2 s0 s39 s0 s101 s15 s3 s111
s5 s159 s3 s6
s1 s5 s60 s60 s7 s42 s6 s4 s11 s1 s2 s30
s0 s7 s0 s6 s0 s2 s49 s1 s17 s5 s5
s0 s51 s26 s1 s2 s113 s2 s29 s0 s55 s0 s9 s0 s12 s66 s0 s2 s41
s52 s12 s1 s14 s8 s0 s18
s8 s10 s37 s6 s0 s3 s144 s38 s0 s53 s2 s46 s0 s76 s0
s157 s0 s55 s5 s46 s86
s1 s2 s2 s40 s4 s2 s20 s1 s0 s28 s39
s0 s1 s128 s89 s19 s0 s2
s54 s47 s6 s1 s0 s68 s140 s1 s74
s7 s138 s11 s0 s5 s22 s0 s1 s0 s11 s4 s9 s2 s22
s9 s79 s24 s0 s0 s151 s0 s2 s126 s0
s110 s119 s0 s155 s16 s0 s8 s0 s12
s77 s14 s84 s47 s76 s0 s31 s1 s2 s3 s15 s19 s108 s0 s1
s90 s87 s111 s1 s0
s2 s159 s97 s3 s3 s0 s99 s1 s0 s85 s16 s24 s34 s0 s0 s8
s53 s23 s2 s22 s0 s23 s14 s17 s6 s1 s33 s7 s37 s3 s0 s11 s18 s102
s1 s0 s7 s1 s51 s56 s117 s64 s21 s1 s58 s33 s37 s28
s1 s0 s21 s7 s0 s31 s1 s45 s8 s11 s6 s25 s151 s50 s0
s31 s11 s0 s1 s0 s1 s1 s5 s2 s3 s56 s0
s78 s40 s9 s0 s1 s4 s3 s0 s5 s6
s117 s14 s2 s6 s9 s9 s1 s1 s3 s45
s29 s0 s81 s1 s34
s2 s1 s4 s9 s24 s136 s69 s5 s6 s5 s1 s121 s0
s57 s0 s106 s62 s0 s15 s12 s27 s2 s4 s1
s10 s16 s11 s0 s152 s132 s0 s116 s10 s0 s57 s5 s31 s59 s11
s59 s0 s2 s8 s68 s2 s58 s1
s1 s0 s4 s0 s7 s28 s21 s139 s83
s18 s3 s0 s1 s46 s19 s50 s0 s91 s4 s105
s1 s42 s9 s23 s35 s16
s59 s21 s2 s18 s79
s14 s10 s35 s3 s8 s7 s23
s3 s64 s4 s149 s88 s22 s61 s11 s34 s0 s149 s51
s0 s2 s94 s154 s68 s0 s41 s6 s0 s8 s103 s79 s4
s29 s152 s0 s23 s37 s38 s31 s0 s0 s43 s17 s1 s0
s3 s28 s45 s62 s17 s66 s22 s105 s89 s0 s2 s4 s1
s5 s1 s76 s47 s68 s4 s2 s1 s76 s101 s3 s78 s4 s6
s32 s1 s0 s80 s111
s117 s1 s0 s128 s72 s16 s56 s15 s93 s1 s6 s24 s1
s17 s2 s9 s7 s13 s102 s54 s5 s56 s72 s44 s12 s90 s8 s0 s2 s22 s4
s35 s7 s30 s3 s4 s9
s147 s56 s1 s1 s10 s14 s80 s132 s96 s3 s14 s0
s7 s97 s5 s62 s5 s3 s2 s1 s5 s0 s60 s138 s0 s0 s0 s108 s0
s3 s82 s4 s23 s52 s4 s6 s19 s0 s10 s0 s1
s28 s54 s0 s73 s28 s1 s99 s2 s48 s0 s6 s9 s88 s63 s11 s0 s5 s4
s87 s0 s7 s0 s10 s0
s90 s104 s1 s102 s4 s25 s122 s15 s14 s0 s15 s1 s4 s4 s137 s20
s40 s38 s60 s8 s1
s27 s29 s6 s34 s7 s11 s2
s0 s0 s20 s6 s0 s79
s57 s66 s1 s1 s39 s59 s1 s1 s1 s0
s131 s80 s0 s42 s58 s0 s13 s5 s18 s83
s133 s72 s4 s43
s5 s7 s7 s94 s100 s3 s2 s22
s12 s1 s0 s5 s0 s8 s34 s2
s36 s3 s21 s25 s95 s77 s6 s27 s1 s0 s31 s94 s37 s9 s19 s34 s13 s3
s1 s95 s66 s32 s0 s0 s103
s44 s1 s0 s15 s61 s2
s95 s134 s98 s5 s37 s67 s1
s0 s4 s4 s10 s2
s48 s2 s1 s0 s17 s111 s81 s15
s42 s65 s79 s0 s1 s7 s133 s18
s20 s59 s0 s17 s2 s159 s3 s0
s30 s55 s21 s3 s111 s0 s52 s5
s0 s0 s6 s0 s2 s1 s2 s6 s2 s3
s0 s158 s0 s17 s1 s1 s8
s1 s14 s5 s0 s1 s14 s4 s41 s7 s79 s40 s8 s3 s1 s2 s3 s3
s7 s0 s16 s0 s11 s0 s70 s0 s22 s29 s2 s15 s3 s1 s1 s0 s24 s0
s2 s101 s38 s21
s4 s1 s2 s1 s0 s3 s1
s1 s1 s73 s3 s0 s1 s43 s2 s12 s62 s17 s39 s140 s2 s79 s12 s1
s9 s1 s1 s0
s9 s44 s0 s11 s48 s0 s11 s6 s5 s11
s85 s54 s3 s13 s1 s1 s1 s0 s1 s1 s24 s1
s0 s0 s1 s2 s71 s64 s110 s4 s0 s18 s90 s0 s60 s0
s6 s24 s0 s10 s62 s0 s7 s1 s0 s8 s6 s20
s148 s47 s0 s0 s2 s1 s0
s1 s48 s42 s137
s2 s105 s115 s0
s21 s0 s83 s52 s2 s27 s121 s0
s2 s71 s6 s57 s0
s133 s3 s44 s1 s119 s1 s0 s0
s10 s5 s25 s1 s64 s31 s0 s2 s3 s107 s6 s17 s107 s4 s117 s2
s3 s67 s50 s84 s0
s2 s3 s19 s14 s45 s53 s2 s0 s36 s26 s2 s0 s0 s1 s2 s1 s0 s0
s146 s4 s2 s3 s0 s0 s0 s2 s146
s17 s99 s73 s36 s15 s1 s38 s5 s29 s1 s114 s0 s48
s0 s0 s63 s5 s1 s46 s1 s137 s2
s2 s3 s4 s2 s0 s1 s1 s24 s4 s32 s6 s3 s0 s1 s91 s6 s2
s1 s42 s8 s0 s18 s153 s0
s11 s95 s16 s2 s23 s3 s12 s0 s13 s7
s58 s8 s141 s72 s0 s20 s72 s2 s29 s1 s1 s3 s2 s129 s1 s5
s0 s2 s1 s0 s3
s62 s0 s5 s15 s58 s0 s24 s3 s0 s3 s20 s0 s67 s0 s1 s3 s0 s104
s1 s141 s122 s6
s42 s0 s0 s3 s79 s95 s30 s16 s0 s106 s22 s17 s65 s1 s18 s26 s39 s0
s0 s62 s9 s64 s4 s43 s0 s7 s3 s1 s2 s5
s0 s60 s0 s3 s154 s114 s7 s29 s4 s67 s2 s1 s0 s19 s12 s92
s7 s45 s13 s89 s14 s79 s73 s30 s3 s8 s0 s33 s21 s56 s0 s0 s21
s3 s1 s75 s1 s48 s69 s133 s0 s0 s0 s2 s94 s20 s0 s0 s38 s18
s68 s126 s32 s3 s4 s0 s1 s44 s8 s70 s24 s13
s24 s109 s22 s0
s0 s3 s33 s1 s44 s1 s5 s2 s0 s82 s3
s26 s1 s63 s14 s1 s26 s19 s13 s7 s42 s34 s15 s0 s1 s141 s4
s41 s45 s11 s22 s7 s2 s9 s19 s123 s56 s0 s20 s0
s24 s24 s0 s1 s3 s1 s4 s4 s3 s0 s74 s125 s17 s2 s6
s1 s6 s6 s1 s8 s39 s0 s8 s10 s1 s20 s1 s13 s13
s0 s25 s44 s3 s7 s23 s2 s4 s13 s2 s0 s95
s5 s7 s0 s128 s99 s0 s12
s114 s146 s42 s2
s0 s0 s37 s0 s17 s56 s13 s0 s1 s11 s123 s22 s28 s15 s3 s16 s0
s44 s12 s9 s6 s1
s20 s0 s39 s14 s0 s79 s15 s7 s117 s1 s0 s86
s121 s8 s20 s2 s28 s33 s15 s73 s7 s0 s97
s22 s5 s1 s2 s41 s31 s1
s5 s10 s50 s3 s21 s23 s10 s17 s7 s17 s7
s46 s26 s33 s102
s31 s30 s9 s34 s0 s56 s134 s91 s54 s5 s109 s12 s1 s11
s21 s24 s12 s36 s4 s111 s13 s3 s20
s82 s6 s2 s131 s4 s1 s3
s71 s6 s24 s2 s14 s23
s10 s2 s46 s5 s21 s46 s33 s0 s0 s57 s21 s0 s0 s8 s1 s107 s1
s6 s17 s40 s2 s59 s39 s14 s13 s4 s134 s4 s1
s6 s39 s45 s124 s15 s0 s0 s20 s11 s1 s0 s3 s2 s3 s3
s10 s0 s73 s0 s5 s149
s60 s9 s106 s30 s109
s7 s24 s0 s1 s81 s0 s0 s57 s25 s7 s71 s1 s0 s55 s20
s1 s3 s45 s109 s0 s0 s2 s2 s37 s25 s37 s4 s5 s115 s0 s0 s0
s14 s1 s39 s0 s0 s2 s9 s19
s0 s3 s82 s133 s49 s1
s2 s0 s140 s1 s1 s3 s1 s1 s104 s0 s62 s11 s0 s0 s22 s12 s1 s2
s0 s44 s2 s9 s2 s2 s79 s43 s0 s43 s104 s29 s27 s1 s28
s115 s1 s25 s6 s1 s0 s17 s1 s14 s5 s5 s84 s0 s12
s0 s11 s13 s8
s92 s138 s7 s2 s2 s91 s6 s1 s62 s9
s36 s4 s3 s28 s3 s42 s34
s0 s3 s9 s0 s4 s43 s42 s0 s2 s2 s26 s0 s147 s0 s10 s29 s119
s0 s14 s141 s47 s117 s11 s0 s4 s44 s0 s23
s14 s16 s0 s18 s4 s5 s1 s15 s24 s86 s65 s142 s139
s4 s4 s4 s0 s2 s2 s21
s0 s5 s2 s3 s0 s4 s40
s0 s7 s58 s7 s142 s12 s82 s0 s19 s9 s3 s91 s0
s1 s116 s58 s3 s52 s122 s37 s36 s63
s2 s32 s0 s71 s97 s40 s54 s15 s53
s55 s1 s52 s4
s1 s1 s5 s47 s9 s62 s0 s2 s0
s1 s14 s0 s1 s0 s30 s2 s0 s57 s6
s12 s4 s27 s10 s3 s0 s2 s6 s3 s4 s21
s49 s32 s33 s3
s1 s135 s5 s18
s10 s0 s45 s124 s0 s9 s1
s62 s7 s40 s1 s74 s1 s151
s49 s39 s4 s0 s64 s44 s62 s57 s1 s18 s7 s4 s5 s24 s158 s7 s3 s19
s2 s4 s138 s47 s3 s0 s4 s4 s87 s11
s0 s43 s22 s0 s4 s4 s6 s24 s0 s0 s28 s0 s65 s4 s0 s42 s1 s158
s16 s6 s0 s57 s90 s130 s20 s14 s1 s130 s35 s24 s28 s65 s3 s0 s152
s9 s13 s5 s13 s86
s0 s89 s0 s2 s48 s0 s130 s108 s0 s20 s104 s11 s11 s15 s12 s14 s25
s9 s1 s7 s2 s58 s0 s0 s91 s0 s12 s30 s3 s106 s3 s1 s0 s90
s0 s21 s87 s24 s117 s2 s3
s3 s13 s14 s33 s22 s0
s0 s2 s13 s0 s27
s1 s104 s111 s31 s82 s42 s22 s37 s96 s1 s34 s1
s7 s114 s12 s0 s11 s0 s0 s8 s7 s0 s3 s2 s2 s6
s0 s0 s26 s3 s78 s1 s109 s1 s2 s3 s1 s5
s7 s5 s0 s0 s71 s0 s74 s4 s0
s11 s4 s2 s1 s0 s4 s0 s18 s2 s42 s112 s84 s36 s45
s29 s1 s27 s13 s27 s29 s47 s7 s13 s1
s68 s1 s78 s30
s0 s43 s13 s13 s39 s55 s30 s0 s11 s62 s57 s119 s10 s10 s0
s1 s23 s29 s0
s24 s1 s48 s1 s0 s0 s16 s2 s13 s2 s6 s76 s103 s0 s119 s9
s42 s54 s32 s46 s1 s16 s11 s95
s112 s0 s23 s16 s23 s49 s0 s36 s0 s2 s67 s13
s1 s133 s98 s10 s7 s4 s38 s1 s3
s17 s55 s0 s2 s35 s1 s65 s2 s23 s0 s99 s0 s5 s4
s56 s1 s9 s6 s0 s0 s1 s6 s12 s37 s96 s0 s137 s5
s123 s39 s41 s5 s7 s91 s1 s9 s34 s3 s7 s0
s134 s1 s38 s32 s35 s2 s1 s17 s155 s4 s1 s9 s1
s5 s10 s6 s0 s133
s1 s1 s70 s8 s0 s16 s73 s100 s112 s75 s0 s9 s30 s20 s3 s0 s87
s92 s5 s122 s29 s0 s109 s0 s1 s13 s113 s0 s2 s0 s0 s5 s14
s92 s0 s64 s71 s21 s4 s11 s0 s37 s12 s14 s30
s4 s3 s31 s63 s60 s1 s0 s24 s34 s46 s92 s3 s0 s4 s2 s0 s25 s9
s149 s2 s6 s5 s67 s0 s8 s7 s0
s155 s139 s80 s10 s57 s35 s2 s38 s2 s5 s0
s37 s9 s19 s0 s5 s22 s6 s33 s0 s3 s0
s21 s5 s3 s40 s12 s18 s2 s113 s22 s19 s40 s97 s0 s1
s46 s82 s2 s127 s7 s26 s24
s1 s1 s61 s1 s1 s37 s70 s115 s0 s3 s17 s3 s36
s73 s1 s1 s0 s3 s4 s1 s0 s6 s0 s8 s0 s0 s0 s66 s56 s146
s38 s18 s18 s110 s156 s0 s4 s116 s0 s3 s41 s0 s15 s101 s11 s0 s0 s5
s61 s3 s66 s59 s94 s0 s35 s2
s16 s3 s129 s2 s108 s3 s42 s21 s45 s1 s159 s8
s2 s4 s1 s0 s122 s4 s17 s0 s71 s38
s14 s8 s120 s0 s0 s2 s157 s28 s16 s3
s5 s125 s1 s2 s1 s2 s2 s61 s142 s4 s75 s0 s8 s58 s0
s4 s3 s23 s74 s2 s95
s12 s8 s1 s58 s7 s0 s0 s9 s13
s3 s7 s0 s10 s2 s10 s16 s0 s0 s31 s32 s149
s7 s20 s2 s127 s44 s0 s2 s90 s37 s51 s1 s81 s22 s5 s26 s1 s89
s47 s24 s11 s6 s22 s61 s0 s2 s1 s54 s2 s6 s9 s125
s101 s1 s7 s68 s11 s1 s4 s19 s33 s2 s1 s45
s2 s0 s21 s3 s0 s3 s80 s0
s10 s5 s100 s4 s1 s16 s14 s47 s12 s0 s8 s1 s1 s43
s4 s62 s88 s13 s1 s129 s7 s3 s0 s115 s27 s0 s1
s112 s0 s0 s3 s10 s60 s35 s12 s1 s9 s6 s0
s0 s9 s54 s1
s3 s1 s120 s45 s3 s1 s0 s0 s36 s0 s47 s4 s26 s17 s0 s39 s79 s0
s15 s1 s0 s0 s46 s7 s12 s19 s3 s22 s6 s0 s10 s8 s27 s127 s0 s17
s0 s53 s3 s1 s7 s21 s63
s5 s105 s39 s0 s1 s0 s30 s1 s6 s0
s2 s10 s84 s0 s51 s42 s6 s44 s40 s1 s133 s113 s1
s0 s94 s27 s2 s0 s3 s45 s90 s0 s104 s1 s0 s0 s0 s1 s43 s7
s2 s55 s8 s10 s0 s118 s1
s2 s146 s13 s57 s4 s112 s18 s30 s0 s5 s0 s3 s13 s2
s70 s0 s4 s0 s4 s51 s61 s0 s1 s29 s0 s0 s75 s0
s11 s0 s0 s3 s1 s82 s9 s9 s17 s8 s1 s35 s35 s95 s8
s6 s42 s148 s0 s20 s64 s56 s101 s21 s67 s4 s15 s12 s6 s6
s0 s79 s10 s76 s95 s0 s5 s9 s16 s112 s0 s16 s8 s0 s26 s33 s0
s40 s1 s2 s8 s3 s26 s0 s99 s115 s0 s8 s3 s3 s1 s11 s52
s42 s0 s10 s120 s19 s113 s118 s41
s8 s1 s1 s31 s95 s0 s2 s143
s0 s48 s2 s0 s48 s13 s1 s4 s74 s30 s2 s1 s13 s0 s1 s52 s33 s0
s18 s86 s139 s1 s115 s72 s15 s0 s128 s37 s4 s2 s121 s1 s57 s6 s28 s58
s3 s0 s52 s25 s1 s10 s4 s57 s2 s1 s44 s39 s29 s0 s0
s0 s0 s70 s1 s96
s0 s26 s1 s9 s1
s32 s0 s0 s52 s5 s9 s66 s31 s10
s60 s4 s4 s67 s39 s2 s57 s3
s31 s55 s3 s4 s1 s3 s18 s24 s30 s11 s0 s21 s67 s77 s0 s4 s8 s18
s0 s0 s29 s23 s5 s0
s0 s2 s30 s18 s107 s2 s5 s35 s1 s1
s3 s1 s58 s46 s101 s7 s47 s54
s47 s6 s12 s38 s100 s0
s25 s0 s13 s13 s3 s4 s16 s46 s45 s66 s10 s1 s34 s0
s41 s5 s16 s1 s18 s0 s0 s1 s48 s0 s5 s154 s1 s0
s79 s31 s60 s155 s4 s0 s0 s7 s27 s61 s0
s25 s2 s100 s15 s0 s2 s1 s39 s1 s12 s6 s3 s74 s9 s2
s4 s109 s108 s0 s20 s72 s0 s36 s14 s9
s27 s0 s0 s132 s12 s1 s0 s7 s2 s0 s20 s1 s118 s1 s85 s3
s138 s25 s43 s0 s4 s37 s0 s45
s0 s0 s9 s3 s3 s2 s48 s2 s0 s2 s1
s24 s2 s56 s94 s1 s4 s62 s9 s37 s2 s56 s62 s44 s80 s4 s9
s0 s14 s3 s4 s0 s27 s82 s36 s0 s2 s1
s41 s28 s0 s0
s153 s49 s4 s13 s1 s0 s8 s0 s1
s26 s3 s5 s1 s9 s88 s7 s3 s7 s5 s5 s9 s32 s31 s0 s114 s3
s0 s12 s2 s73 s1 s0 s4 s40 s35 s3 s34 s0 s6 s79 s0 s1 s3 s1
s1 s136 s1 s1 s116 s40 s68 s113 s125 s4 s1 s12
s6 s0 s1 s5 s1 s8 s19 s10 s0 s0
s6 s0 s4 s57
s0 s46 s94 s0 s39 s0 s98 s7 s8 s96
s10 s35 s49 s116 s137
s16 s9 s8 s49 s0 s3 s0 s22 s5 s88
s123 s0 s1 s9 s38 s121 s137 s106 s45 s6 s2 s73 s38
s0 s2 s2 s152 s69 s0 s7 s51 s4
s0 s22 s135 s19 s138 s58 s18 s4 s1 s125 s114 s90 s3 s38
s8 s0 s8 s0 s0 s73
s143 s149 s1 s4 s8 s7 s0 s4 s48 s9 s23 s27 s18
s3 s2 s4 s73 s48 s0 s0 s13 s0 s54 s4 s143 s3 s11
s45 s57 s53 s85 s40 s1 s9 s29
s12 s24 s1 s13 s32 s146 s5 s3 s3 s12 s2 s5 s20 s35 s33 s22
s9 s2 s147 s63 s48 s75 s1 s10 s17
s0 s16 s0 s8 s27 s34
s29 s94 s30 s1 s0 s130 s71 s5
s0 s19 s8 s0 s61 s9 s12 s38 s2 s4 s0
s2 s1 s3 s106 s81 s2 s0 s0 s6 s0 s26 s16 s5
s31 s37 s55 s2 s0 s40 s26 s6 s5 s9 s123 s0 s15
s0 s10 s0 s2 s93 s10 s2 s39 s0 s36 s53
s11 s96 s23 s21 s0 s0 s20 s46 s73
s9 s5 s0 s0 s1
s110 s84 s3 s58 s56 s30 s5 s3
s0 s0 s3 s122 s3
s138 s5 s3 s26 s38 s25 s2 s11 s5 s39 s4 s103 s5 s1 s75 s0
s52 s97 s1 s39 s2 s121 s103
s15 s36 s21 s28 s1 s33 s37 s8 s11 s5
s6 s60 s7 s10 s28 s0 s0 s4 s0 s1 s0 s0 s67
s3 s1 s126 s44 s97 s14 s0 s37 s3 s21 s6 s13 s0 s90 s2 s72
s9 s42 s0 s18 s16 s0
s8 s4 s28 s0 s54 s23 s34 s0 s31 s9
s10 s43 s23 s38 s3 s29 s0 s0 s9
s5 s2 s2 s0 s43 s13 s2 s31 s7 s0 s52 s53 s12 s21 s12
s143 s0 s54 s1 s0 s49 s88
s17 s2 s3 s17 s38 s36 s12 s141 s151 s107 s2 s36 s0 s152 s13 s0 s0
s32 s5 s9 s13 s0 s4 s62 s149 s8 s2
s6 s8 s3 s54 s147 s6 s6 s1 s2 s4
s2 s14 s0 s6 s8 s0 s1 s0 s10 s20 s0 s13 s73 s0 s15 s2
s2 s9 s0 s2
s145 s39 s18 s0 s2 s0 s3 s0 s0 s0 s6 s0 s0
s0 s2 s4 s2 s5 s2 s1 s3 s24 s18 s142 s0 s53 s0 s1 s0 s8 s0
s16 s1 s1 s18 s37 s136 s26 s20
s59 s5 s61 s0 s2 s153 s0
s12 s1 s11 s123 s1 s12 s9 s0 s17 s1 s94 s70 s10 s1 s0 s1 s10
s1 s32 s0 s8 s0 s32
s1 s94 s17 s152 s45 s23 s7 s0 s58 s3 s0 s36 s147 s35 s0 s3
s33 s1 s4 s0 s90 s97 s0 s1 s5 s18
s1 s134 s2 s31 s30 s11 s64
s26 s1 s2 s97
s1 s27 s0 s85 s43 s65 s141 s21 s82 s157 s0 s0 s37
s47 s37 s4 s2 s128 s0 s48 s2 s11 s29 s98 s43 s0 s74 s136 s0 s2
s14 s131 s9 s7 s72
s69 s5 s2 s1
s6 s6 s3 s12 s36 s29 s33 s10 s61 s0 s14 s61 s35 s125 s0 s22 s47 s4
s9 s31 s8 s10 s7 s0 s19 s0
s18 s6 s57 s1 s0
s35 s142 s118 s1 s4 s37 s0 s58 s22 s2
s24 s22 s2 s113 s13 s65 s0 s9 s1 s5 s3
s110 s97 s0 s5 s4 s12 s21 s5 s23 s18 s8 s0 s1 s4 s31
s0 s0 s6 s136
s15 s0 s95 s13 s52 s0 s56 s5 s3 s80 s6 s2 s4 s15 s1 s9 s3 s0
s77 s0 s22 s30 s10 s9 s25 s0 s130 s17 s7 s2 s2 s17 s1 s29 s0 s4
s3 s25 s1 s128 s16 s0 s1 s72 s0 s0 s1 s0 s1 s8 s60 s6 s12 s0
s20 s0 s3 s51 s0 s0 s2 s1 s60 s80
s85 s7 s112 s7 s34 s27 s0 s6
s82 s0 s14 s40 s6 s8 s0 s1 s1 s6 s0 s0 s15 s147 s1
s71 s3 s24 s0 s3 s0 s0 s2 s3 s0 s20 s20 s2
s43 s34 s45 s140 s16 s118 s147 s15 s43
s33 s4 s54 s1 s0
s3 s22 s143 s49 s2 s1 s104 s1 s31 s20 s13 s2 s138 s43
s1 s1 s1 s133 s7 s3 s0 s4 s52 s64 s121 s1 s0 s1 s0 s74
s1 s30 s2 s16 s0 s2 s42 s61 s12 s14 s0 s15 s14 s1 s33
s0 s125 s0 s7 s8 s7 s153 s0 s53
s45 s40 s1 s4 s26 s9 s46 s1 s5 s3
s10 s9 s19 s33 s11
s0 s147 s45 s41 s11
s140 s1 s1 s68 s9 s1 s136 s63 s4 s19 s34 s3
s3 s0 s65 s0 s1 s1 s5 s138
s80 s78 s0 s14 s0 s80 s9 s0 s55 s24 s0 s71 s0 s88 s158 s6
s0 s104 s27 s0 s2 s66 s72 s27 s120 s46 s106 s4 s6 s3 s1 s2 s0 s118
s58 s14 s7 s0 s24 s37 s15 s40 s8 s1 s120 s0 s2 s1 s17 s7 s13 s7
s2 s2 s48 s2 s54 s3 s26 s0 s11 s6 s46 s17 s22 s0 s72
s10 s158 s9 s73 s64 s0 s8 s6 s56 s29 s19 s88 s0
s1 s26 s9 s44 s0 s25 s80 s0
s1 s12 s8 s108
s8 s36 s3 s1 s3 s1 s9 s24 s1 s29 s9 s49 s41
s61 s11 s19 s20 s6 s4 s141 s0 s75 s0 s0 s43 s0 s23 s1 s71 s13
s54 s79 s91 s0
s6 s4 s2 s5 s1 s71 s99 s6 s0 s2 s26 s142 s39 s0 s3 s0 s12
s0 s36 s17 s19 s2
s13 s0 s53 s0 s130 s22 s119 s2 s82 s2 s0 s39 s108 s103 s0
s0 s1 s6 s0 s105 s0 s109 s49 s12 s10 s50 s3
s1 s1 s37 s48 s11 s18 s19
s0 s1 s40 s30 s4 s51 s27 s6 s8 s1
s1 s59 s51 s154 s49 s16 s1 s39 s12 s5 s14 s0 s109 s4 s107 s45
s12 s0 s141 s21 s63 s4 s3 s5 s0 s28 s11 s17 s2 s0 s23 s62
s1 s2 s2 s2 s0 s47
s21 s3 s19 s18 s1 s95 s31 s13 s18 s21
s29 s0 s54 s65 s2 s1 s0
s3 s12 s11 s2 s18 s1 s24 s2 s1 s3 s1 s50 s40 s0 s89 s57 s0
s0 s0 s0 s64 s6 s7 s24 s1
s23 s1 s158 s3 s41 s34 s6 s24 s2 s34 s1 s0 s4 s29 s31 s2
s2 s80 s39 s27 s0 s4 s48 s12 s0 s68 s100 s41 s15 s0 s3 s6 s13
s106 s3 s47 s0 s27 s0 s9 s84 s2
s14 s19 s40 s0 s97 s0 s110 s1 s8
s6 s2 s6 s38 s45 s0 s136 s10 s92 s0 s72 s15 s0 s0 s1 s98
s11 s0 s4 s68 s66 s1 s43 s3 s1 s13 s4 s44 s85
s38 s95 s78 s85 s88 s133 s67 s18 s2 s158
s12 s68 s115 s9 s39 s62 s21 s148
s61 s7 s104 s13 s2 s4 s2 s115 s22 s11 s99 s16 s9 s6 s75
s2 s0 s9 s18 s1 s15 s0 s71 s3 s0 s2 s86 s1 s1
s0 s9 s60 s10 s0 s18 s44
s22 s1 s13 s5 s14 s80 s8 s16 s15 s1 s63 s9
s0 s120 s7 s100 s0 s7 s27 s0 s12 s129
s43 s51 s67 s157
s44 s7 s47 s52 s25 s1 s129 s8 s33 s2
s1 s3 s29 s13 s1 s95 s110 s54 s2 s0 s21 s23 s88
s12 s9 s0 s49 s0 s1 s2 s110 s96 s3 s4
s2 s20 s5 s17 s2 s0 s41 s15
s5 s40 s8 s0 s4 s1
s1 s8 s54 s1 s0 s1 s32 s0 s15 s135 s85
s30 s81 s83 s61
s1 s9 s6 s115 s4 s95 s30 s101 s119 s77 s7 s25 s0 s121 s68 s7 s53 s19
s156 s4 s66 s27 s45 s26 s0 s111 s17 s135 s108 s0 s2
s136 s3 s1 s7 s1 s0 s7 s6 s0 s126 s153 s58 s2 s20
s62 s33 s0 s152 s5 s134 s0 s41 s81 s63 s21 s157 s1 s6
s23 s0 s5 s154 s4 s0 s25 s70 s0 s18 s0 s49 s17 s58 s118 s37
s3 s18 s10 s0 s48 s3
s68 s2 s5 s37 s74 s2 s25 s4 s7 s8
s7 s10 s0 s44 s4 s7 s22 s76 s0 s1 s19 s85
s0 s149 s123 s0 s25
s60 s4 s2 s7 s0 s0 s0 s30 s2 s55 s119
s0 s3 s13 s14 s4 s5 s2 s0 s3
s41 s0 s0 s13 s43 s1 s149 s7 s0 s16 s158 s147 s138 s6 s80
s18 s96 s104 s3 s22 s0 s23 s1 s0 s43 s156 s28 s46 s109
s54 s76 s0 s143 s3 s0 s2 s21 s4 s15 s48 s2 s1
s2 s9 s0 s23 s0 s9 s0 s16 s0 s1 s44 s0
s2 s98 s13 s28 s9 s75 s41 s0
s0 s25 s12 s8 s0 s37 s0 s28 s52 s1 s74 s1 s4 s53 s83
s12 s74 s7 s2 s0 s0 s118 s11 s96 s0 s0 s80 s0 s7 s22
s2 s0 s4 s133 s37 s0 s0 s11 s20 s106
s21 s40 s2 s0 s133 s109 s0 s0 s62 s22 s49 s8 s1 s1 s95 s11 s10 s17
s6 s12 s2 s3 s48 s42 s0 s8 s43 s0 s6
